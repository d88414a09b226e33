#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dither or trajectory or split_phase" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
for nt in 128 64; do SNES_DITHER_NT=$nt python bench.py --config dither --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/d_$nt.json 2> $O/d_$nt.err; python -c "
import json
d=json.loads(open('$O/d_$nt.json').read().strip().splitlines()[-1]); print('dither NT $nt', round(d['value']), d['ms_per_step'])"; done
python bench.py --config perceptual --steps 60 --no-cpu-baseline --no-extras > $O/p.json 2>$O/p.err; python -c "
import json
d=json.loads(open('$O/p.json').read().strip().splitlines()[-1]); print('perceptual', round(d['value']), d['ms_per_step'])"
python bench.py --config images --steps 40 > $O/i.json 2>$O/i.err; python -c "
import json
d=json.loads(open('$O/i.json').read().strip().splitlines()[-1]); print('images', round(d['value']), d['ms_per_step'], 'init', d['config']['init_seconds'])"
