#!/bin/bash
# profiles/r4_images_ab.sh: throughput mode and the reference's loop, the library in the tree against profiles/ab/lib*.so, interleaved
cd $GRAFT_REPO_ROOT
L=snesimage_amd/libsnesimage_hip.so
cp $L /tmp/libtree.so
for i in 1 2; do
  for f in profiles/ab/lib*.so /tmp/libtree.so; do
    TAG=$(basename $f .so | sed 's/^lib//'); cp $f $L
    python bench.py --config images --steps 40 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s images: value %.0f ms/step %.4f' % ('$TAG', d['value'], d['ms_per_step']))"
  done
done
cp /tmp/libtree.so $L
bash profiles/r4_slots_ab.sh
