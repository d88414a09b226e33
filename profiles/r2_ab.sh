#!/bin/bash
# same-box A/B: libsnesimage_hip.so (new) against libsnesimage_hip_base.so (previous build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc = 0 ] || exit 1
cp snesimage_amd/libsnesimage_hip.so $O/new.so
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', round(d['value']), '%.3f ms' % d['ms_per_step'], 'ref64', round(d['reference_batch']['value']), 'v2 us', round(r.get('kernel_us', 0),1) if 'kernel_us' in r else r)"; }
for rep in 1 2; do
for v in new base; do
  if [ $v = base ]; then cp snesimage_amd/libsnesimage_hip_base.so snesimage_amd/libsnesimage_hip.so; else cp $O/new.so snesimage_amd/libsnesimage_hip.so; fi
  for cfg in ${CFGS:-rgb}; do st=200; [ $cfg = dither ] && st=40
    python bench.py --config $cfg --steps $st --no-cpu-baseline > $O/${v}_${cfg}_$rep.json 2> $O/${v}_${cfg}_$rep.err || { tail -5 $O/${v}_${cfg}_$rep.err; exit 1; }
    show $O/${v}_${cfg}_$rep.json "$v $cfg #$rep"
  done
done; done
cp $O/new.so snesimage_amd/libsnesimage_hip.so
