#!/bin/bash
# profiles/r4_pmc_sq.sh NAME CPL SCRIPT [args...]: where the waves of every kernel spend their cycles (separate --pmc passes, kernel
# trace only).  WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_* = issuing; all in quad-cycles
# summed over waves (MI355X_MICROARCH.md, PMC slots).  -> gpurun_out/NAME/sq.json + a table on stdout
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1; CPL=$2; SCRIPT=$3; shift 3
mkdir -p $O
ARGS="$*"
run() { n=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/$O/$n -o p -f csv -- python3 $GRAFT_REPO_ROOT/$SCRIPT $ARGS > $GRAFT_REPO_ROOT/$O/$n.log 2>&1 )
}
( cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1 ) || true
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES || echo 'pass failed: run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES'
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT || echo 'pass failed: run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT'
run mem FETCH_SIZE || echo 'pass failed: run mem FETCH_SIZE'
run memw WRITE_SIZE || echo 'pass failed: run memw WRITE_SIZE'
run l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum || echo 'pass failed: run l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum'
python profiles/pmc_summary.py $O/sq.json $CPL "rocprofv3 --kernel-trace --pmc <one pass per counter set> -- python3 $SCRIPT $ARGS" $O/sq1 $O/sq2 $O/mem $O/memw $O/l2 > /dev/null
f=$(find $O/sq1 -name '*kernel_trace.csv' | head -1)
python profiles/r4_sq_table.py $O/sq.json $f | tee $O/sq_table.txt
rm -rf $O/sq1 $O/sq2 $O/mem $O/memw $O/l2
