#!/bin/bash
# throughput under env-variable sweeps: usage r2_sweep.sh OUTDIR "<VAR=a,b,c>" [more bench args]
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
spec=$2; shift 2
var=${spec%%=*}; vals=${spec#*=}
for v in ${vals//,/ }; do
  for l in 1 2; do
    env $var=$v SNES_LANES=$l python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $O/b_${v}_l$l.json 2> $O/b_${v}_l$l.err
    python -c "
import json
d=json.loads(open('$O/b_${v}_l$l.json').read().strip().splitlines()[-1])
print('$var=$v lanes $l', round(d['value']), 'ms/step %.3f' % d['ms_per_step'], 'H %.3f V %.3f' % (d['roofline'].get('h_ms', 0), d['roofline']['avg_launch_ms']))"
  done
done
