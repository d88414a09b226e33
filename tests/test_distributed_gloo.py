"""N > 1 path on CPU: two `gloo` ranks shard the candidates of every optimizer call, exchange one
min-all-reduce and must end bit-identical to the single-process run.  The sharding / collective /
commit sequencing under test is snesimage_amd.distributed.sharded_step (the code bench.py runs
with the HIP scorer over RCCL); here the scorer is backed by the CPU oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleShardScorer:
    """begin/commit protocol of HipShardScorer implemented over the CPU oracle."""

    def __init__(self, image, O):
        self.o, self.O = image, O

    def begin(self, method, palette, index, channel, seed, step_id, n_total, rank, world):
        o = self.o
        cur = o.palette[palette * o.sub_size + index].copy()
        if method == 0:
            cand = self.O.random_candidates(seed, step_id, n_total or 64)
        elif method == 1:
            cand = np.repeat(cur[None, :], 32, 0)
            cand[:, channel] = np.arange(32)
        else:
            cand = np.stack([self.O.nes_color(i) for i in range(56)])
        self.cand, self.slot, self.method = cand, (palette, index), method
        self.incumbent = o.error() if method != 2 else 1.7976931348623157e308
        errs = torch.full((len(cand),), float("inf"), dtype=torch.float64)
        own = np.arange(rank, len(cand), world)
        if len(own):
            errs[own] = torch.from_numpy(o.score_candidates(palette, index, cand[own]))
        return errs

    def commit(self, errors):
        best, best_k = self.incumbent, -1
        for k, e in enumerate(errors.tolist()):  # ascending k, strict < (lib.rs:216-219)
            if e < best:
                best, best_k = e, k
        if self.method == 2 and best_k < 0:
            best_k = 0
        if best_k >= 0:
            pal = self.o.palette
            pal[self.slot[0] * self.o.sub_size + self.slot[1]] = self.cand[best_k]
            self.o.palette = pal
        self.o.optimize()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, calls, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from snesimage_amd.distributed import sharded_step
    from snesimage_amd.synth import synth_image
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img = synth_image(0x5EED0002, 256, 64)
    o = O.OracleImage(img, 2, 3)
    o.initialize_tiles()
    o.recalculate_palettes()
    sc = OracleShardScorer(o, O)
    for i, (method, p, idx, ch, _) in enumerate(calls):
        sharded_step(sc, method, p, idx, ch, 1, i, 12 if method == 0 else 0)
    out_q.put((rank, o.palette.tolist(), float(o.error()).hex(), o.palette_map.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_steps_match_single_process(O):
    from snesimage_amd.distributed import sharded_step
    from snesimage_amd.synth import synth_image
    sched = O.schedule(2, 3, 30)
    calls = [sched[0], sched[1], sched[24], sched[25]]  # two random calls, two channel calls
    # single process (world = 1): the same code path without a process group
    img = synth_image(0x5EED0002, 256, 64)
    o = O.OracleImage(img, 2, 3)
    o.initialize_tiles()
    o.recalculate_palettes()
    sc = OracleShardScorer(o, O)
    for i, (method, p, idx, ch, _) in enumerate(calls):
        sharded_step(sc, method, p, idx, ch, 1, i, 12 if method == 0 else 0)
    # ... which itself equals the oracle's own step() (the reference's loop)
    o2 = O.OracleImage(img, 2, 3)
    o2.initialize_tiles()
    o2.recalculate_palettes()
    for i, (method, p, idx, ch, _) in enumerate(calls):
        o2.step(method, p, idx, ch, 1, i, 12 if method == 0 else 0)
    assert o.palette.tolist() == o2.palette.tolist() and o.error() == o2.error()
    want = (o.palette.tolist(), float(o.error()).hex(), o.palette_map.tobytes())

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, calls, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, pal, err, mp_bytes in got:
        assert (pal, err, mp_bytes) == want, "rank %d diverged" % rank
