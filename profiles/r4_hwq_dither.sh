#!/bin/bash
# plain --dither / --perceptual-palettes runs against the queue count after the lanes' streams became lazy
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
run() { python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG $* %.4f ms/step %.3f M/s hwq=%s' % (d['ms_per_step'], d['value']/1e6, d.get('hw_queues')))" | tee -a $O/log7.txt; }
TAG="plain default" run --config dither --steps 30
for q in 3 5 8; do TAG="plain hwq$q" GPU_MAX_HW_QUEUES=$q run --config dither --steps 30; done
TAG="plain default" run --config perceptual --steps 40
for q in 5 8; do TAG="plain hwq$q" GPU_MAX_HW_QUEUES=$q run --config perceptual --steps 40; done
TAG="plain default" run --steps 100
