#!/bin/bash
# profiles/r3_ab.sh NAME [bench args...]: headline step twice (noise), kernel stats once -> gpurun_out/NAME/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; shift
O=gpurun_out/$n; mkdir -p $O
for i in 1 2; do python bench.py --no-extras --no-cpu-baseline --steps 300 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f ms/step %.4f | V %.3f ms group %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['group_ms']))"; done
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
f=$(find $O/kt -name '*.db' | head -1); python profiles/dbstats.py $f 14 | cut -c1-60,70-140
rm -rf $O/kt
