#!/bin/bash
# address-translation counters of the bench's kernels (one pass) -> gpurun_out/r2tlb/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2tlb; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum -d $GRAFT_REPO_ROOT/$O/a -o p -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/a.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum -d $GRAFT_REPO_ROOT/$O/b -o p -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/b.log 2>&1 )
python profiles/pmc_summary.py $O/tlb.json 4096 "rocprofv3 --kernel-trace --pmc TCP_UTCL1_* -- python3 bench.py --steps 6 --warmup 2" $O/a $O/b | head -8
