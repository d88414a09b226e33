#!/bin/bash
# round-2 baseline: tests, bench, standalone (1-lane) and 2-lane kernel stats
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_rgb4096.json 2> $O/bench_rgb4096.err
python bench.py --steps 400 --warmup 10 --batch 64 --no-cpu-baseline > $O/bench_rgb64.json 2> $O/bench_rgb64.err
SNES_LANES=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_rgb4096_l1.json 2> $O/bench_rgb4096_l1.err
cd /tmp
SNES_LANES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_l1 -o l1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_l1.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_l2 -o l2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_l2.log 2>&1
cd $GRAFT_REPO_ROOT
for d in l1 l2; do f=$(find $O/prof_$d -name '*.db' | head -1); [ -n "$f" ] && python profiles/dbstats.py $f 30 > $O/stats_$d.txt; done
tail -3 $O/pytest.log; cat $O/bench_rgb4096.json | cut -c1-400
