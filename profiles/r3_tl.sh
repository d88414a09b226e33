#!/bin/bash
# profiles/r3_tl.sh NAME ARGS... : kernel trace of profiles/r3_slots.py ARGS; per-kernel stats and the timeline of one window
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; shift
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/profiles/r3_slots.py "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 )
f=$(find $O/kt -name '*.db' | head -1)
python profiles/dbstats.py $f 40 > $O/kernel_stats.txt
python profiles/dbtimeline.py $f kw_commit 3 > $O/timeline.txt
tail -2 $O/run.log
cat $O/kernel_stats.txt
cat $O/timeline.txt
rm -rf $O/kt
