#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -4 $O/pytest.log
for l in 1 2; do SNES_LANES=$l python bench.py --config dither --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_dither_l$l.json 2> $O/bench_dither_l$l.err; python -c "
import json
d=json.loads(open('$O/bench_dither_l$l.json').read().strip().splitlines()[-1]); print('dither lanes $l', round(d['value']), d['ms_per_step'])"; done
