#!/bin/bash
# kernel-trace stats of a few configurations -> gpurun_out/r2s/ (scratch; r2_final.sh produces the committed evidence)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
stats() { # name, bench args
  n=$1; shift 1
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/kt_$n.log 2>&1 )
  f=$(find $O/kt_$n -name '*.db' | head -1); python profiles/dbstats.py $f 30 > $O/kernel_stats_$n.txt; rm -rf $O/kt_$n
}
for h in 1024 2048 4096 8192; do SNES_HGRID=$h python bench.py --steps 150 --no-cpu-baseline --no-extras > $O/hgrid_$h.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/hgrid_$h.json').read().strip().splitlines()[-1]); print('hgrid $h', round(d['value']), '%.3f' % d['ms_per_step'])"; done
stats rgb
stats rgb_batch64 --batch 64 --steps 200
stats perceptual --config perceptual
stats dither --config dither --steps 10 --warmup 2
head -14 $O/kernel_stats_rgb.txt
