#!/bin/bash
# kernel-trace profile of the default bench with one launch lane (standalone kernel durations) -> gpurun_out/$1/stats.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
cd /tmp
SNES_LANES=${2:-1} rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline ${@:3} > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name '*.db' | head -1); python profiles/dbstats.py $f 30 > $O/stats.txt
head -16 $O/stats.txt
