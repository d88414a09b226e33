#!/bin/bash
# the one-rank RCCL run of bench.py against hardware-queue count and stream arrangement
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
run() { python bench.py --steps 100 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG %.4f ms/step %.3f M/s' % (d['ms_per_step'], d['value']/1e6))" | tee -a $O/log2.txt; }
for i in 1 2; do
TAG="plain" run
TAG="plain hwq8" GPU_MAX_HW_QUEUES=8 run
export SNES_BENCH_FORCE_DIST=1
TAG="dist" run
TAG="dist hwq8" GPU_MAX_HW_QUEUES=8 run
TAG="dist hwq16" GPU_MAX_HW_QUEUES=16 run
TAG="dist v0_aside=0" SNES_V0_ASIDE=0 run
TAG="dist h0_min=0" SNES_H0_MIN=0 run
TAG="dist base_stream=0" SNES_BASE_STREAM=0 run
TAG="dist hwq8 h0_min=0" GPU_MAX_HW_QUEUES=8 SNES_H0_MIN=0 run
unset SNES_BENCH_FORCE_DIST
done
