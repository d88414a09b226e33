#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
for h in 8192 16384 32768 65536; do SNES_HGRID=$h python bench.py --steps 200 --no-cpu-baseline --no-extras > $O/hgrid_$h.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/hgrid_$h.json').read().strip().splitlines()[-1]); print('hgrid $h', round(d['value']), '%.3f' % d['ms_per_step'])"; done
