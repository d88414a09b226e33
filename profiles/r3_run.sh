#!/bin/bash
# profiles/r3_run.sh OUTDIR -- run GPU steps one after the other; a step that times out or is killed ends the call
# (no further GPU step is started behind a hang).  Usage on the GPU box: bash profiles/r3_run.sh r3a 'cmd1' 'cmd2' ...
out=gpurun_out/$1; shift
mkdir -p "$out"
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "== step $i: $cmd" | tee -a "$out/steps.log"
  timeout -k 10 ${STEP_TIMEOUT:-800} bash -c "$cmd" > "$out/step$i.log" 2>&1
  rc=$?
  echo "== step $i rc=$rc" | tee -a "$out/steps.log"
  tail -n ${TAIL:-25} "$out/step$i.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "step $i timed out or crashed: stopping"; exit $rc; fi
done
exit 0
