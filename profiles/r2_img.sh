#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
python -m pytest tests/test_throughput.py -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
for g in 2 4 8; do python bench.py --config images --steps 40 --warmup 5 --groups $g > $O/img_g$g.json 2> $O/img_g$g.err; python -c "
import json
d=json.loads(open('$O/img_g$g.json').read().strip().splitlines()[-1]); print('images groups $g', round(d['value']), 'ms/step %.3f' % d['ms_per_step'], 'init %.2f s' % d['config']['init_seconds'])"; done
