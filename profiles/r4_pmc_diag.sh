#!/bin/bash
# profiles/r4_pmc_diag.sh NAME: the counter tool over the slot windows, variant by variant (each its own process; a host-side
# SIGSEGV of one variant is the finding, not a GPU fault: the next variant still runs) -> gpurun_out/NAME/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
run() { # tag, env..., -- extra script args
  tag=$1; shift
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/kt_$tag -o p -f csv -- python3 $GRAFT_REPO_ROOT/profiles/r4_pmc_diag.py $O/$tag $EXTRA > $O/$tag.log 2>&1 ); rc=$?
  echo "== $tag rc=$rc : $(grep -c '^DIAG' $O/$tag.log) markers, last: $(grep '^DIAG' $O/$tag.log | tail -1)"
  rm -rf $O/kt_$tag
}
EXTRA="" run default X=1
EXTRA="" run no_window_side SNES_WINDOW_SIDE=0
EXTRA="" run depth1 SNES_WINDOW_DEPTH=1
EXTRA="" run no_base_stream SNES_BASE_STREAM=0
EXTRA="--steps-first" run steps_first X=1
EXTRA="" run hwq8 GPU_MAX_HW_QUEUES=8
exit 0
