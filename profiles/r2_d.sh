#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
for l in 2 3 4; do SNES_LANES=$l python bench.py --config dither --steps 40 --no-cpu-baseline --no-extras > $O/lanes_$l.json 2>$O/lanes_$l.err; python -c "
import json; d=json.loads(open('$O/lanes_$l.json').read().strip().splitlines()[-1]); print('dither lanes $l', round(d['value']), '%.3f' % d['ms_per_step'])"; done
for l in 2 3; do SNES_LANES=$l python bench.py --config perceptual --steps 100 --no-cpu-baseline --no-extras > $O/plan_$l.json 2>$O/plan_$l.err; python -c "
import json; d=json.loads(open('$O/plan_$l.json').read().strip().splitlines()[-1]); print('perceptual lanes $l', round(d['value']), '%.3f' % d['ms_per_step'])"; done
