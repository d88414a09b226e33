"""bench.py --gpus N started plainly (no torchrun environment) spawns its own ranks: snesimage_amd/launch.py.
Covered on CPU with two gloo ranks running a stand-in for the benchmark body."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
s = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(s)
dist.barrier()
if rank == 0:
    print(json.dumps({"world": world, "min": t.item(), "sum": s.item(), "args": sys.argv[1:], "local_rank": os.environ["LOCAL_RANK"]}), flush=True)
else:
    print("noise from rank", rank, flush=True)  # must not reach the job's stdout
dist.destroy_process_group()
sys.exit(3 if (rank == 1 and "--fail" in sys.argv) else 0)
"""


def test_needs_spawn_only_without_a_launcher_environment():
    from snesimage_amd.launch import needs_spawn
    assert needs_spawn(2, {}) and needs_spawn(8, {"PATH": "x"})
    assert not needs_spawn(1, {})
    assert not needs_spawn(2, {"WORLD_SIZE": "2", "RANK": "0"})  # started by torch.distributed.run: the ranks exist already


def test_spawn_ranks_runs_one_process_per_rank_and_relays_rank0(tmp_path):
    from snesimage_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    code, out = spawn_ranks(2, [str(child), "--steps", "3"], timeout=300)
    assert code == 0
    assert "noise from rank" not in out  # rank 0's stdout only
    lines = [l for l in out.splitlines() if l.startswith("{")]  # (gloo prints a connection banner on stdout)
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"world": 2, "min": 1.0, "sum": 3.0, "args": ["--steps", "3"], "local_rank": "0"}
    code, _ = spawn_ranks(2, [str(child), "--fail"], timeout=300)
    assert code == 3  # a failing rank fails the job


def test_a_rank_that_dies_before_the_rendezvous_ends_the_job(tmp_path, capfd):
    """Rank 1 exits before init_process_group: rank 0 would wait in the rendezvous for ever.  The launcher watches every
    rank, names the one that failed on stderr, ends the others by pid and returns the failing rank's code — promptly."""
    import time
    from snesimage_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text("import os, sys\nif os.environ['RANK'] == '1':\n    sys.exit(7)\n" + CHILD)
    t0 = time.monotonic()
    code, out = spawn_ranks(2, [str(child)], timeout=600)
    dt = time.monotonic() - t0
    assert code == 7
    assert dt < 120, "the job must end when the rank dies, not at the rendezvous' own timeout (took %.0f s)" % dt
    assert not [l for l in out.splitlines() if l.startswith("{")]  # rank 0 never got through the rendezvous
    err = capfd.readouterr().err
    assert "rank 1 of 2 exited with code 7" in err


def test_job_timeout_ends_every_rank(tmp_path, capfd):
    from snesimage_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text("import time\ntime.sleep(600)\n")
    code, _ = spawn_ranks(2, [str(child)], timeout=2)
    assert code == 124
    assert "still running" in capfd.readouterr().err


def test_bench_spawns_before_touching_the_gpu():
    """bench.py consults the launcher before it imports torch or the library (a process that has initialised HIP must not
    be turned into ranks)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("needs_spawn(") < main.index("import torch")


def test_bench_asks_for_more_hardware_queues_for_the_rgb_configuration_only():
    """GPU_MAX_HW_QUEUES is read when the HIP runtime starts: bench.py decides before it imports torch, for the configuration whose
    launch groups run on three streams (profiles/r4_hw_queues*.txt), never over a value the caller has set, and the child
    runs of the other configurations do not inherit it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    w = bench.wants_more_hw_queues
    assert w("rgb", {}) and w("rgb", {"WORLD_SIZE": "8"}) and w("rgb", {"SNES_BENCH_FORCE_DIST": "1"})
    assert not w("dither", {"WORLD_SIZE": "8"}) and not w("perceptual", {}) and not w("images", {"WORLD_SIZE": "8"})
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("wants_more_hw_queues(") < main.index("import torch")
    assert '"GPU_MAX_HW_QUEUES" not in os.environ' in main
    extras = src[src.index("def config_extras("):src.index("def cpu_baseline(")]
    assert "_SET_HW_QUEUES and k == \"GPU_MAX_HW_QUEUES\"" in extras and "env=env" in extras
