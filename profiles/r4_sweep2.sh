#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f ms/step %.4f | V %.3f ms group %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['group_ms']))"; }
for i in 1 2; do
run X=1
run SNES_DOWN1_GRID=16384
run SNES_DOWN1_GRID=8192
run SNES_DOWN1_GRID=4096
run SNES_DOWN1_GRID=2048
run SNES_DOWN1=0
run SNES_LPT=0
done
