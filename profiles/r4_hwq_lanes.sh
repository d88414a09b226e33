#!/bin/bash
# do streams that exist but never run (the second launch lane's) decide how the active ones share hardware queues?  plain run, queue count x lanes
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
run() { python bench.py --steps 100 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG $* %.4f ms/step %.3f M/s' % (d['ms_per_step'], d['value']/1e6))" | tee -a $O/log5.txt; }
for l in 2 1; do for q in 4 5 8; do TAG="plain lanes=$l hwq$q" SNES_LANES=$l GPU_MAX_HW_QUEUES=$q run; done; done
export SNES_BENCH_FORCE_DIST=1
for l in 2 1; do for q in 4 8; do TAG="dist lanes=$l hwq$q" SNES_LANES=$l GPU_MAX_HW_QUEUES=$q run; done; done
