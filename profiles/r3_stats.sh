#!/bin/bash
# profiles/r3_stats.sh NAME SCRIPT [args]: rocprofv3 --kernel-trace --stats of one run -> gpurun_out/NAME/kernel_stats.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=$1; sc=$2; shift 2
O=gpurun_out/$n; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o p -- python3 $GRAFT_REPO_ROOT/$sc "$@" > $GRAFT_REPO_ROOT/$O/run.log 2>&1 ) || exit 1
f=$(find $O/kt -name '*.db' | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python3 $sc $*" > $O/kernel_stats.txt
python profiles/dbstats.py $f 34 >> $O/kernel_stats.txt
rm -rf $O/kt
head -16 $O/kernel_stats.txt
