#!/bin/bash
# where the one-rank RCCL run of bench.py (SNES_BENCH_FORCE_DIST=1) loses 0.3 ms per step against the plain run: both under the kernel trace, timelines around a commit
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/dist_gap; mkdir -p $O
for v in 0 1; do
  export SNES_BENCH_FORCE_DIST=$v
  python bench.py --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force_dist=$v %.4f ms/step %.3f M/s' % (d['ms_per_step'], d['value']/1e6))" | tee -a $O/log.txt
done
export SNES_BENCH_FORCE_DIST=1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 ) || exit 1
f=$(find $O/kt -name '*.db' | head -1); python profiles/dbstats.py $f 30 > $O/kernel_stats_dist.txt; python profiles/dbtimeline.py $f k_commit 30 > $O/timeline_dist.txt; rm -rf $O/kt
tail -50 $O/timeline_dist.txt | cut -c1-150
