#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
python bench.py --steps 200 --no-cpu-baseline --no-extras > $O/bench.json 2>$O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('rgb', round(d['value']), d['ms_per_step'], 'V', d['roofline']['avg_launch_ms'])"
