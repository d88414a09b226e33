#!/bin/bash
# PMC passes of one bench configuration (counters in their own runs: --kernel-trace + --pmc only)
# usage: r2_pmc.sh OUTDIR LANES CANDIDATES_PER_LAUNCH <bench args...>
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; L=$2; CPL=$3; shift 3
mkdir -p $O
run() { # name, counters...
  n=$1; shift
  ( cd /tmp && [ "$L" != default ] && export SNES_LANES=$L; cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/$n -o p -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > $GRAFT_REPO_ROOT/$O/$n.log 2>&1 )
}
BENCH_ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run valu VALUBusy VALUUtilization SQ_INSTS_VALU SQ_WAVES
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
run l2 TCC_HIT_sum TCC_MISS_sum
python profiles/pmc_summary.py $O/pmc.json $CPL "rocprofv3 --kernel-trace --pmc <one pass per counter set> -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS (SNES_LANES=$L)" $O/fetch $O/write $O/valu $O/wait $O/l2
