#!/bin/bash
# profiles/r3_tl_bench.sh NAME [bench args]: timeline of one headline step (kernels between two k_commit launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; shift
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 )
f=$(find $O/kt -name '*.db' | head -1)
python profiles/dbtimeline.py $f k_commit ${TLN:-3} > $O/timeline.txt
cat $O/timeline.txt
rm -rf $O/kt
