#!/bin/bash
# lanes x chunk sweep of the default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for l in 2 3 4; do for ch in 683 1024 1366 2048; do
  SNES_LANES=$l SNES_CHUNK=$ch python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/b_l${l}_c$ch.json 2> $O/b_l${l}_c$ch.err
  python -c "
import json
d=json.loads(open('$O/b_l${l}_c$ch.json').read().strip().splitlines()[-1])
print('lanes $l chunk $ch', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
done; done
