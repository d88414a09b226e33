#!/bin/bash
# profiles/r4_tlw.sh NAME WINDOW [slots args]: timeline of one slot window of the given size right after the k-means start -> gpurun_out/NAME/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; w=$2; shift 2
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && SNES_WINDOW_DEPTH=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/profiles/r4_slots.py --converge 0 --calls 240 --window $w "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 )
f=$(find $O/kt -name '*.db' | head -1)
python profiles/dbtimeline.py $f kw_commit 3 > $O/timeline.txt
tail -1 $O/run.log | cut -c1-400
cat $O/timeline.txt | cut -c1-110
rm -rf $O/kt
