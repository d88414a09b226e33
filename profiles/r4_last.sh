#!/bin/bash
# the round's last GPU call: --dither / --perceptual-palettes with their lanes' streams eager again, then profiles/r4_refresh.sh
# (GPU tests, the counter passes bench.py reads, kernel statistics and the default invocation on this build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
run() { python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* %.4f ms/step %.3f M/s hwq=%s %s' % (d['ms_per_step'], d['value']/1e6, d.get('hw_queues'), d['library'][-12:]))" | tee -a $O/last_check.txt; }
run --config dither --steps 30
run --config perceptual --steps 40
bash profiles/r4_refresh.sh
