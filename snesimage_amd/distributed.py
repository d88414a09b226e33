"""Candidate sharding of one optimizer step across ranks.

One process per GPU.  The candidates of a step are independent given the incumbent palette
(lib.rs:205-220), so rank r scores candidates k with k % world == r and the only exchange is ONE
min-all-reduce per step over an n_total-element float64 vector (+inf where a rank does not own
the candidate; errors are finite and >= 0).  Every rank then applies the reference's acceptance
rule (ascending k, strict <, lib.rs:216-219) to the identical reduced vector, so the palettes stay
bit-identical without a broadcast.  `torch.distributed` carries the collective: backend "nccl" is
RCCL over xGMI on MI355X; "gloo" is used by the CPU tests.

A *scorer* is any object with
    begin(method, palette, index, channel, seed, step_id, n_total, rank, world) -> 1-D float64 torch
        tensor of the step's n candidates (this rank's entries filled in, +inf elsewhere), and
    commit(errors) -> None
`HipShardScorer` is the product scorer (the C-ABI split-phase step).
"""
import torch
import torch.distributed as dist

from . import api


class HipShardScorer:
    """Split-phase step over libsnesimage_hip.so; the error vector lives in HBM."""

    def __init__(self, image: "api.OptimizedImage", device: torch.device):
        self.image = image
        self.device = device
        self._buf = None
        # The library and the collective share one explicit stream: `sharded_step` issues the all-reduce with this stream
        # current, so RCCL is ordered behind the scoring kernels and the commit behind RCCL.  (torch's default stream has
        # handle 0, which the library reads as "keep your own stream" — never hand it that one.)
        self.stream = torch.cuda.Stream(device)
        self.image.set_stream(self.stream.cuda_stream)

    def begin(self, method, palette, index, channel, seed, step_id, n_total, rank, world):
        n = n_total if method == api.METHOD_RANDOM else (32 if method == api.METHOD_CHANNEL else 56)
        if self._buf is None or self._buf.numel() != n:
            self._buf = torch.empty(n, dtype=torch.float64, device=self.device)
        self.image.step_begin(method, palette, index, channel, seed, step_id, n_total, rank, world,
                              self._buf.data_ptr())
        return self._buf

    def commit(self, errors):
        self.image.step_commit(errors.data_ptr())


def sharded_step(scorer, method, palette, index, channel, seed, step_id, n_total, group=None):
    """One optimizer call with its candidates sharded over the ranks of `group`."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    errors = scorer.begin(method, palette, index, channel, seed, step_id, n_total, rank, world)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        stream = getattr(scorer, "stream", None)  # device scorers: the stream their kernels run on
        if stream is not None:
            with torch.cuda.stream(stream):
                dist.all_reduce(errors, op=dist.ReduceOp.MIN, group=group)
        else:
            dist.all_reduce(errors, op=dist.ReduceOp.MIN, group=group)
    scorer.commit(errors)
    return errors
