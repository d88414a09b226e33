#!/bin/bash
# profiles/r4_kt.sh NAME [bench args]: kernel statistics and the timeline of one step of `bench.py --no-extras --no-cpu-baseline` -> gpurun_out/NAME/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; shift
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 100 "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 )
f=$(find $O/kt -name '*.db' | head -1)
python profiles/dbstats.py $f 16 > $O/stats.txt
python profiles/dbtimeline.py $f k_commit 3 > $O/timeline.txt
cat $O/stats.txt | cut -c1-150
cat $O/timeline.txt | cut -c1-120
rm -rf $O/kt
