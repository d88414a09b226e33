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


# ---- slot windows: the reference's own loop (64 / 32 / 56 candidates per call), calls dealt to the ranks -----------------
#
# `sharded_step` splits ONE call's candidates; its base image is computed by every rank.  A *window* (DESIGN.md,
# snesimage_run_slots) holds K consecutive calls of the schedule, all scored against the same palette: the calls are dealt to the
# ranks in runs of six consecutive calls, round robin — every call its own base image, its own candidates — and the only exchange is ONE min-all-reduce over the
# window's K x n error vector.  Every rank then applies the calls' decisions in order and stops behind the first call that
# changed the palette (lib.rs:216-219: strict <), so the ranks stay bit-identical without a broadcast and the trajectory is
# the reference's, call for call.
#
# A *window scorer* has
#     slots_begin(n_slots, seed, first_step_id, state, rank, world) -> (errors, taken, stride): 1-D float64 tensor of at
#         least taken * stride entries, this rank's calls filled in and +inf elsewhere (fewer calls than asked for where the
#         method changes: all calls of a window have `stride` candidates), and
#     slots_commit(errors, taken) -> (consumed, accepted, log): calls that took effect, whether the last one changed the
#         palette, and per consumed call (error, best_k, rgb5, changed).


def schedule_advance(sub_count, sub_size, state, n, nes=False):
    """Scheduler state (palette, index, channel, step) after n more calls (lib.rs:917-932)."""
    import ctypes as C

    from . import _ffi
    L = _ffi.load()
    p, i, ch, st, m = (C.c_uint32(int(v)) for v in (*state, 0))
    for _ in range(n):
        L.snesimage_schedule_next(sub_count, sub_size, int(nes), C.byref(p), C.byref(i), C.byref(ch), C.byref(st), C.byref(m))
    return (p.value, i.value, ch.value, st.value)


class HipWindowScorer(HipShardScorer):
    """Slot windows over libsnesimage_hip.so; the error vector lives in HBM."""

    MAX_WINDOW = 1024  # calls per window over all ranks (the library takes at most SNES_WINDOW_MAX, default 64, per rank)

    def __init__(self, image, device):
        super().__init__(image, device)
        self._wbuf = torch.empty(self.MAX_WINDOW * 64, dtype=torch.float64, device=device)

    def slots_begin(self, n_slots, seed, first_step_id, state, rank, world):
        taken, stride = self.image.slots_begin(min(n_slots, self.MAX_WINDOW), seed, first_step_id, state, 0, rank, world,
                                               self._wbuf.data_ptr())
        return self._wbuf, taken, stride

    def slots_commit(self, errors, taken):
        return self.image.slots_commit(errors.data_ptr(), taken)


class WindowPolicy:
    """Calls per window (the library's own rule in snesimage_run_slots): with acceptance rate p a window of K calls gets
    (1 - (1 - p)^K) / p of them through before the first acceptance voids the rest and costs about t0 + t1 K / world
    (t0 / t1 ~ 15); p from the recent windows, older ones fading by 0.85 per window.  Speed only: results do not depend on it."""

    def __init__(self, world=1, lo=8, hi=64):
        self.world, self.lo, self.hi = world, lo * world, hi * world
        self.calls, self.accepts, self.k = 0.0, 0.0, self.lo

    def update(self, taken, consumed, accepted):
        self.calls = 0.85 * self.calls + consumed
        self.accepts = 0.85 * self.accepts + accepted
        p = (self.accepts + 0.5) / (self.calls + 8.0)
        best, k = -1.0, self.lo
        while k <= self.hi:
            rate = (1.0 - (1.0 - p) ** k) / p / (15.0 + k / self.world)
            if rate > best:
                best, self.k = rate, k
            k += (4 if k < 32 * self.world else 8) * self.world
        if not accepted and consumed == taken:  # a clean window: at least twice as many next
            self.k = max(self.k, min(2 * taken, self.hi))


def sharded_run_slots(scorer, sub_count, sub_size, n_calls, seed=1, first_step_id=0, state=(0, 0, 0, 0), window=0, group=None, nes=False):
    """The reference's loop for n_calls calls, the calls of every window dealt to the ranks of `group`.
    Returns (log, state, stats) like OptimizedImage.run_slots."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    policy = WindowPolicy(world)
    done, log = 0, []
    stats = {"calls": 0, "accepted": 0, "windows": 0, "scored": 0, "useful": 0}
    while done < n_calls:
        k = min(window or policy.k, n_calls - done)
        errors, taken, stride = scorer.slots_begin(k, seed, first_step_id + done, state, rank, world)
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            vec = errors[: taken * stride]  # contiguous view: reduced in place
            stream = getattr(scorer, "stream", None)
            if stream is not None:
                with torch.cuda.stream(stream):
                    dist.all_reduce(vec, op=dist.ReduceOp.MIN, group=group)
            else:
                dist.all_reduce(vec, op=dist.ReduceOp.MIN, group=group)
        consumed, accepted, wlog = scorer.slots_commit(errors, taken)
        log += wlog or []
        state = schedule_advance(sub_count, sub_size, state, consumed, nes)
        done += consumed
        stats["calls"] += consumed
        stats["accepted"] += accepted
        stats["windows"] += 1
        stats["scored"] += taken * stride
        stats["useful"] += consumed * stride
        policy.update(taken, consumed, accepted)
    return log, state, stats
