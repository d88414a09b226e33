#!/bin/bash
# profiles/r4_ab_env.sh "VAR=a VAR=b ...": the headline step under each environment setting (one per word; "-" = none), interleaved three times
cd $GRAFT_REPO_ROOT
one() { python bench.py --no-extras --no-cpu-baseline --steps 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s value %.0f ms/step %.4f | V %.3f ms' % ('$TAG', d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for i in 1 2 3; do
  for w in $1; do
    TAG=$w
    if [ "$w" = "-" ]; then one; else env $(echo $w | tr ',' ' ') bash -c "$(declare -f one); TAG=$w one"; fi
  done
done
