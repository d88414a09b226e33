#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
for q in 4 8 16; do for cfg in rgb perceptual dither; do st=100; [ $cfg = dither ] && st=30
GPU_MAX_HW_QUEUES=$q python bench.py --config $cfg --steps $st --no-cpu-baseline > $O/q_${cfg}_$q.json 2> $O/q_${cfg}_$q.err; python -c "
import json
d=json.loads(open('$O/q_${cfg}_$q.json').read().strip().splitlines()[-1]); print('$cfg queues $q', round(d['value']), '%.3f' % d['ms_per_step'], 'ref64', round(d['reference_batch']['value']))"; done; done
