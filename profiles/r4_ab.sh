#!/bin/bash
# profiles/r4_ab.sh NAME: the quick A/B of a kernel change: headline step twice + its timeline, the 64-candidate call's timeline -> gpurun_out/NAME/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; shift
O=gpurun_out/$n; mkdir -p $O
for i in 1 2; do python bench.py --no-extras --no-cpu-baseline --steps 300 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f ms/step %.4f | V %.3f ms group %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['group_ms']))"; done
python bench.py --no-extras --no-cpu-baseline --steps 400 --batch 64 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 64: value %.0f ms/step %.4f' % (d['value'], d['ms_per_step']))"
TLN=3 bash profiles/r3_tl_bench.sh ${n}_tl4096 "$@" > $O/tl4096.txt 2>&1; cat $O/tl4096.txt | cut -c1-100
TLN=3 bash profiles/r3_tl_bench.sh ${n}_tl64 --batch 64 --steps 100 "$@" > $O/tl64.txt 2>&1; cat $O/tl64.txt | cut -c1-100
