#!/bin/bash
# after the launch lanes' streams became lazy: GPU tests, smoke, and the queue-count matrix again (plain and one-rank RCCL; default lanes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
run() { python bench.py --steps 100 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG $* %.4f ms/step %.3f M/s hwq=%s' % (d['ms_per_step'], d['value']/1e6, d.get('hw_queues')))" | tee -a $O/log6.txt; }
for q in 3 4 5 8; do TAG="plain hwq$q" GPU_MAX_HW_QUEUES=$q run; done
TAG="plain default" run
export SNES_BENCH_FORCE_DIST=1
for q in 4 5 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run; done
TAG="dist default" run
for q in 4 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run --config perceptual --steps 40; done
for q in 4 8; do TAG="dist hwq$q" GPU_MAX_HW_QUEUES=$q run --config dither --steps 30; done
