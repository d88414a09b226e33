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


# ---- slot windows: the calls of a window dealt to the ranks (snesimage_amd.distributed.sharded_run_slots) -------------------
class OracleWindowScorer:
    """slots_begin / slots_commit protocol of HipWindowScorer over the CPU oracle: K consecutive calls of the schedule scored
    against the current palette (runs of six consecutive calls dealt round robin), then committed in order up to the first call that accepts."""

    def __init__(self, image, O, count, size):
        self.o, self.O, self.count, self.size = image, O, count, size

    def slots_begin(self, n_slots, seed, first_step_id, state, rank, world):
        from snesimage_amd.distributed import schedule_advance
        o, calls, st = self.o, [], tuple(state)
        while len(calls) < n_slots:
            method = 0 if st[3] % 5 < 4 else 1  # lib.rs:890
            if calls and method != calls[0][0]:
                break
            calls.append((method, st[0], st[1], st[2]))
            st = schedule_advance(self.count, self.size, st, 1)
        n = 64 if calls[0][0] == 0 else 32
        errs = torch.full((len(calls) * n,), float("inf"), dtype=torch.float64)
        self.cands = []
        for j, (method, p, i, ch) in enumerate(calls):
            if method == 0:
                cand = self.O.random_candidates(seed, first_step_id + j, 64)
            else:
                cand = np.repeat(o.palette[p * o.sub_size + i][None, :], 32, 0)
                cand[:, ch] = np.arange(32)
            self.cands.append(cand)
            if (j // 6) % world == rank:  # runs of six consecutive calls, round robin
                errs[j * n:(j + 1) * n] = torch.from_numpy(o.score_candidates(p, i, cand))
        self.calls, self.n, self.incumbent = calls, n, o.error()
        return errs, len(calls), n

    def slots_commit(self, errors, taken):
        log, e = [], errors.tolist()
        for j in range(taken):
            best, best_k = self.incumbent, -1
            for k in range(self.n):  # ascending k, strict < (lib.rs:216-219)
                if e[j * self.n + k] < best:
                    best, best_k = e[j * self.n + k], k
            _, p, i, _ = self.calls[j]
            if best_k >= 0:
                pal = self.o.palette
                pal[p * self.o.sub_size + i] = self.cands[j][best_k]
                self.o.palette = pal
                self.o.optimize()
                log.append((best, best_k, self.cands[j][best_k].copy(), 1))
                return j + 1, 1, log
            log.append((self.incumbent, -1, self.o.palette[p * self.o.sub_size + i].copy(), 0))
        return taken, 0, log


WINDOW_FIRST, WINDOW_CALLS = 18, 16  # 2 x 3 entries: calls 18..23 are the last random calls of step 3, 24.. the channel sweeps of step 4


def _window_worker(rank, world, port, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from snesimage_amd.distributed import schedule_advance, sharded_run_slots
    from snesimage_amd.synth import synth_image
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = O.OracleImage(synth_image(0x5EED0002, 256, 64), 2, 3)
    o.initialize_tiles()
    o.recalculate_palettes()
    log, state, stats = sharded_run_slots(OracleWindowScorer(o, O, 2, 3), 2, 3, WINDOW_CALLS, seed=1, first_step_id=WINDOW_FIRST,
                                          state=schedule_advance(2, 3, (0, 0, 0, 0), WINDOW_FIRST), window=5)
    out_q.put((rank, o.palette.tolist(), float(o.error()).hex(), o.palette_map.tobytes(), [(float(e).hex(), int(k), c.tolist(), ch) for e, k, c, ch in log], state, stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_slot_windows_match_the_call_by_call_loop(O):
    """Windows of up to five calls over two gloo ranks against the oracle's own step() per scheduled call: same decisions
    call for call, same final palette, error and palette_map on both ranks."""
    from snesimage_amd.synth import synth_image
    sched = O.schedule(2, 3, WINDOW_FIRST + WINDOW_CALLS + 1)
    ref = O.OracleImage(synth_image(0x5EED0002, 256, 64), 2, 3)
    ref.initialize_tiles()
    ref.recalculate_palettes()
    want_log = []
    for j in range(WINDOW_FIRST, WINDOW_FIRST + WINDOW_CALLS):
        method, p, idx, ch, _ = sched[j]
        before = ref.palette.copy()
        e, best = ref.step(method, p, idx, ch, 1, j, 0)
        want_log.append((float(e).hex(), best.tolist(), int(not np.array_equal(before, ref.palette))))
    want = (ref.palette.tolist(), float(ref.error()).hex(), ref.palette_map.tobytes())
    assert {m for m, *_ in sched[WINDOW_FIRST:WINDOW_FIRST + WINDOW_CALLS]} == {0, 1}  # random and channel calls

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_window_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=800) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, pal, err, mp_bytes, log, state, stats in got:
        assert (pal, err, mp_bytes) == want, "rank %d diverged" % rank
        assert [(e, c, ch) for e, _, c, ch in log] == want_log
        m, p, i, ch, st = sched[WINDOW_FIRST + WINDOW_CALLS]
        assert tuple(state) == (p, i, ch, st)
        assert stats["calls"] == WINDOW_CALLS and stats["useful"] <= stats["scored"]
