#!/bin/bash
# profiles/r4_ab_w1.sh: the reference's loop — call by call (windows of one call) and in adaptive windows — with the windows' side stream
# borrowed from the parent context or its own, and with / without the second candidate stream
cd $GRAFT_REPO_ROOT
run() { python profiles/r4_slots.py --converge 30 --calls 960 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['from_kmeans_start']; b=d['converged']
print('%-26s start %6.0f calls/s %.3f M useful | converged %6.0f calls/s %.3f M useful' % ('$TAG', a['calls_per_s'], a['useful_cand_per_s']/1e6, b['calls_per_s'], b['useful_cand_per_s']/1e6))"; }
for i in 1 2; do
  for v in "SNES_WINDOW_BORROW=1" "SNES_WINDOW_BORROW=0" "SNES_WINDOW_BORROW=1,SNES_WINDOW_AUX=0"; do
    TAG="w1 $v" env $(echo $v | tr ',' ' ') bash -c "$(declare -f run); TAG='w1 $v' run --window 1"
    TAG="adaptive $v" env $(echo $v | tr ',' ' ') bash -c "$(declare -f run); TAG='ad $v' run"
  done
done
