#!/bin/bash
# same-box A/B: launch lanes for each configuration
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
for cfg in rgb perceptual dither; do for l in 1 2; do
  st=100; [ $cfg = dither ] && st=30
  SNES_LANES=$l python bench.py --config $cfg --steps $st --no-cpu-baseline --no-extras > $O/${cfg}_$l.json 2> $O/${cfg}_$l.err; python -c "
import json
d=json.loads(open('$O/${cfg}_$l.json').read().strip().splitlines()[-1]); print('$cfg lanes $l', round(d['value']), '%.3f' % d['ms_per_step'])"; done; done
