#!/bin/bash
# profiles/r4_sweep.sh: the reference's loop from the k-means start under a few threshold settings -> stdout
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python profiles/r4_slots.py --converge 0 --calls 1920 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['from_kmeans_start']
print('%.3f M useful/s, %.0f calls/s, %d windows' % (a['useful_cand_per_s']/1e6, a['calls_per_s'], a['windows']))"; }
for i in 1 2; do
run X=1
run SNES_H2Q_MAX=256
run SNES_H2Q_MAX=1024
run SNES_H2Q_MAX=2048
run SNES_SCAN4_MAX=8192
run SNES_SCAN4_MAX=0
run SNES_WINDOW_SIDE=0
done
