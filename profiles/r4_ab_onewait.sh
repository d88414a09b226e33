#!/bin/bash
# profiles/r4_ab_onewait.sh: a short list's V passes behind one wait for B's sweeps (SNES_ONE_WAIT=1) or one wait each (0) — the knob existed in the
# build this script measured (commit 8c8b6f9's parent working tree) and was removed with the experiment (-0.4 %, not kept):
# the 64-candidate call, the 32-candidate channel calls, the reference's loop call by call and in adaptive windows.  Interleaved, same box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_onewait; mkdir -p $O
b64() { python bench.py --batch 64 --steps 600 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG batch64 %.4f ms/call' % d['ms_per_step'])"; }
slots() { python profiles/r4_slots.py --converge 30 --calls 960 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['from_kmeans_start']; b=d['converged']
print('$TAG slots $* start %6.0f calls/s %.3f M useful | converged %6.0f calls/s %.3f M useful' % (a['calls_per_s'], a['useful_cand_per_s']/1e6, b['calls_per_s'], b['useful_cand_per_s']/1e6))"; }
for i in 1 2 3; do
  for v in 0 1; do
    export SNES_ONE_WAIT=$v; TAG="one_wait=$v"
    b64 | tee -a $O/log.txt
    slots --window 1 | tee -a $O/log.txt
    slots | tee -a $O/log.txt
  done
done
# the headline and the memory the context holds (lane storage allocated for the lanes in use)
unset SNES_ONE_WAIT
python bench.py --steps 200 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.4f ms/step %.3f M/s hbm_in_use_gb %s' % (d['ms_per_step'], d['value']/1e6, d.get('hbm_in_use_gb')))" | tee -a $O/log.txt
python bench.py --config perceptual --steps 50 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('perceptual %.4f ms/step %.3f M/s hbm_in_use_gb %s' % (d['ms_per_step'], d['value']/1e6, d.get('hbm_in_use_gb')))" | tee -a $O/log.txt
echo done
