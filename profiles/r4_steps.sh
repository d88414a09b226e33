#!/bin/bash
# profiles/r4_steps.sh NAME marker "substr ..." SCRIPT [args]: per-step intervals of a traced run -> gpurun_out/NAME/steps.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; marker=$2; subs=$3; sc=$4; shift 4
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/$sc "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 ) || exit 1
f=$(find $O/kt -name '*.db' | head -1)
python profiles/dbsteps.py $f $marker $subs > $O/steps.txt
rm -rf $O/kt
tail -3 $O/steps.txt
