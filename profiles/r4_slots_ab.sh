#!/bin/bash
# profiles/r4_slots_ab.sh: the reference's loop through the slot windows, the library in the tree against profiles/ab/lib*.so, same box,
# interleaved: from the k-means start (adaptive windows) and converged -> stdout (useful cand/s)
cd $GRAFT_REPO_ROOT
L=snesimage_amd/libsnesimage_hip.so
cp $L /tmp/libtree.so
for i in 1 2; do
  for f in profiles/ab/lib*.so /tmp/libtree.so; do
    TAG=$(basename $f .so | sed 's/^lib//'); cp $f $L
    python profiles/r4_slots.py --converge 30 --calls 1920 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['from_kmeans_start']; b=d.get('converged')
print('%-10s start: %.3f M useful/s (acc %.3f, %d windows, %.0f calls/s) | converged: %.3f M useful/s (acc %.4f, %.0f calls/s)' % ('$TAG', a['useful_cand_per_s']/1e6, a['acceptance'], a['windows'], a['calls_per_s'], b['useful_cand_per_s']/1e6, b['acceptance'], b['calls_per_s']))"
  done
done
cp /tmp/libtree.so $L
