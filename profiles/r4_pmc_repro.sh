#!/bin/bash
# profiles/r4_pmc_repro.sh NAME: is the round-3 SIGSEGV under `rocprofv3 --pmc` the tool's or the library's?  (each step its own process)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
R=$GRAFT_REPO_ROOT/profiles/micro/pmc_queue_repro
st() { echo "== $1 rc=$2 : $(tail -1 $O/$1.log | cut -c1-160)"; }
( cd /tmp && $R 300000 > $O/plain.log 2>&1 ); st plain $?
( cd /tmp && rocprofv3 --kernel-trace -d $O/kt_a -o p -f csv -- $R 300000 > $O/trace_only.log 2>&1 ); st trace_only $?; rm -rf $O/kt_a
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/kt_b -o p -f csv -- $R 300000 1000 > $O/pmc_sync1000.log 2>&1 ); st pmc_sync1000 $?; rm -rf $O/kt_b
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/kt_c -o p -f csv -- $R 300000 > $O/pmc_nosync.log 2>&1 ); st pmc_nosync $?; rm -rf $O/kt_c
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/kt_d -o p -f csv -- python3 $GRAFT_REPO_ROOT/profiles/r4_slots.py --converge 100 --calls 480 --window 64 --sync-every 60 > $O/slots_sync60.log 2>&1 ); st slots_sync60 $?; rm -rf $O/kt_d
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/kt_e -o p -f csv -- python3 $GRAFT_REPO_ROOT/profiles/r4_slots.py --converge 100 --calls 480 --window 64 > $O/slots_nosync.log 2>&1 ); st slots_nosync $?; rm -rf $O/kt_e
exit 0
