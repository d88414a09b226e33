#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dither or trajectory or split_phase or golden" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do python bench.py --config dither --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/d_$i.json 2> $O/d_$i.err; python -c "
import json
d=json.loads(open('$O/d_$i.json').read().strip().splitlines()[-1]); print('dither', round(d['value']), d['ms_per_step'])"; done
