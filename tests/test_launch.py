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


def test_bench_spawns_before_touching_the_gpu():
    """bench.py consults the launcher before it imports torch or the library (a process that has initialised HIP must not
    be turned into ranks)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("needs_spawn(") < main.index("import torch")
