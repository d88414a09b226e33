#!/bin/bash
# round-2 run c: maps diet — parity tests + bench (1 and 2 lanes)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for l in 1 2; do
  SNES_LANES=$l python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_l$l.json 2> $O/bench_l$l.err
  python -c "
import json
d=json.loads(open('$O/bench_l$l.json').read().strip().splitlines()[-1])
print('lanes $l', round(d['value']), 'ms/step %.3f' % d['ms_per_step'], 'V ms %.3f' % d['roofline']['avg_launch_ms'])"
done
python bench.py --config dither --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dither.json 2> $O/bench_dither.err; python -c "
import json
d=json.loads(open('$O/bench_dither.json').read().strip().splitlines()[-1]); print('dither', round(d['value']))"
