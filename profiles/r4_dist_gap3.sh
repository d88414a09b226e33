#!/bin/bash
# GPU_MAX_HW_QUEUES against the plain run and the one-rank RCCL run of bench.py (headline and the 64-candidate call)
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
run() { python bench.py --steps 100 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG $* %.4f ms/step %.3f M/s' % (d['ms_per_step'], d['value']/1e6))" | tee -a $O/log3.txt; }
for q in 2 3 4 5 6 8; do TAG="plain hwq$q" GPU_MAX_HW_QUEUES=$q run; done
for q in 2 4 6; do TAG="plain hwq$q" GPU_MAX_HW_QUEUES=$q run --batch 64 --steps 300; done
export SNES_BENCH_FORCE_DIST=1
for q in 4 5 6 8 12; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run; done
for q in 4 6 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run --batch 512; done
for q in 4 6 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run --config dither --steps 30; done
for q in 4 6 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run --config perceptual --steps 40; done
