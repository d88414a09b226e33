#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
for i in 1 2; do python bench.py --steps 150 --no-cpu-baseline > $O/rgb$i.json 2> $O/rgb$i.err; python -c "
import json
d=json.loads(open('$O/rgb$i.json').read().strip().splitlines()[-1]); print('rgb', round(d['value']), '%.3f' % d['ms_per_step'], 'ref64', round(d['reference_batch']['value']), d['roofline']['avg_launch_ms'], d['roofline']['group_ms'])"; done
SNES_BASE_STREAM=0 python bench.py --steps 150 --no-cpu-baseline > $O/rgb3.json 2> $O/rgb3.err; python -c "
import json
d=json.loads(open('$O/rgb3.json').read().strip().splitlines()[-1]); print('rgb nobase', round(d['value']), '%.3f' % d['ms_per_step'], 'ref64', round(d['reference_batch']['value']))"
