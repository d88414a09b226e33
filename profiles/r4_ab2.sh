#!/bin/bash
# profiles/r4_ab2.sh [bench args]: the library in the tree ("tree") against every profiles/ab/lib*.so (builds of other commits or
# variants), same box, runs interleaved three times: headline step and the 64-candidate call -> stdout
cd $GRAFT_REPO_ROOT
L=snesimage_amd/libsnesimage_hip.so
cp $L /tmp/libtree.so
one() { python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s n=%-5d value %.0f ms/step %.4f | V %.3f ms' % ('$TAG', d['config']['batch'], d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for i in 1 2 3; do
  for f in profiles/ab/lib*.so /tmp/libtree.so; do
    TAG=$(basename $f .so | sed 's/^lib//'); cp $f $L
    one --steps 300 "$@"; one --steps 600 --batch 64 "$@"
  done
done
cp /tmp/libtree.so $L
