#!/bin/bash
# timeline of one optimizer step per configuration -> gpurun_out/r2tl/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2tl; mkdir -p $O
tl() { n=$1; shift 1
  ( cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/kt_$n.log 2>&1 )
  f=$(find $O/kt_$n -name '*.db' | head -1); python profiles/dbtimeline.py $f k_commit 2 > $O/timeline_$n.txt; rm -rf $O/kt_$n; }
tl dither --config dither
tl rgb
tl rgb64 --batch 64
tl perceptual --config perceptual
cat $O/timeline_dither.txt
