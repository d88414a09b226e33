#!/bin/bash
# round-2 run b: new tests + H-pass grid cap sweep (1 lane and 2 lanes)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for hg in 2048 4096 8192 16384; do
  for l in 1 2; do
    SNES_HGRID=$hg SNES_LANES=$l python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_h${hg}_l$l.json 2> $O/bench_h${hg}_l$l.err
    python -c "
import json
d=json.loads(open('$O/bench_h${hg}_l$l.json').read().strip().splitlines()[-1])
print('hgrid $hg lanes $l', round(d['value']), 'ms/step %.3f' % d['ms_per_step'])"
  done
done
