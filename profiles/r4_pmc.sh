#!/bin/bash
# PMC passes of one workload (counters in their own runs: --kernel-trace + --pmc only, one counter set per pass)
# usage: r4_pmc.sh OUTDIR CANDIDATES_PER_LAUNCH SCRIPT [script args...]     (SCRIPT relative to the repo root, run with python3)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; CPL=$2; SCRIPT=$3; shift 3
mkdir -p $O
run() { # name, counters...
  n=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/$n -o p -f csv -- python3 $GRAFT_REPO_ROOT/$SCRIPT $ARGS > $GRAFT_REPO_ROOT/$O/$n.log 2>&1 )
}
ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run valu VALUBusy VALUUtilization SQ_INSTS_VALU SQ_WAVES
run l2 TCC_HIT_sum TCC_MISS_sum
python profiles/pmc_summary.py $O/pmc.json $CPL "rocprofv3 --kernel-trace --pmc <one pass per counter set> -- python3 $SCRIPT $ARGS" $O/fetch $O/write $O/valu $O/l2
rm -rf $O/fetch $O/write $O/valu $O/l2
