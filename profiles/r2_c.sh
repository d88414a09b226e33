#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
for rep in 1 2; do for l in 1 0; do SNES_LPT=$l python bench.py --batch 64 --steps 2000 --no-cpu-baseline --no-extras > $O/lpt_$l.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/lpt_$l.json').read().strip().splitlines()[-1]); print('batch64 lpt $l', round(d['value']), '%.4f ms' % d['ms_per_step'])"; done; done
