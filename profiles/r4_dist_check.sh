#!/bin/bash
# the one-rank RCCL run as bench.py now sets it up (GPU_MAX_HW_QUEUES=8 for the RGB configuration with a process group on RCCL) beside the plain run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
SNES_BENCH_FORCE_DIST=1 python bench.py --no-config-extras --no-cpu-baseline --steps 100 > $O/bench_rgb_rccl_one_rank.json 2> $O/bench_rgb_rccl_one_rank.err || exit 1
python bench.py --no-extras --no-cpu-baseline --steps 100 > $O/bench_rgb_plain_check.json 2>/dev/null || exit 1
python3 -c "
import json
for f in ('bench_rgb_rccl_one_rank','bench_rgb_plain_check'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, '%.4f ms/step %.3f M/s' % (d['ms_per_step'], d['value']/1e6), d.get('hw_queues'), d.get('reference_batch',{}).get('value'))"
