#!/bin/bash
# round-4 evidence on the final build, part 3 of profiles/r4_final.sh (the other configurations' bench lines, the reference's loop, the
# one-GPU sharding proxies) and a rehearsal of `python bench.py --gpus 2` / `--gpus 4` as the driver types it (ranks sharing the one device,
# gloo standing in for RCCL: code paths, not a measurement) -> gpurun_out/r4final/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
bash profiles/r4_final.sh 3 || exit 1
echo "== rehearsal --gpus 2 / 4"
for n in 2 4; do
  SNES_BENCH_SHARE_GPU=1 SNES_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus $n --steps 20 --warmup 3 --no-cpu-baseline > $O/rehearsal_gpus$n.json 2> $O/rehearsal_gpus$n.err || { tail -5 $O/rehearsal_gpus$n.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$O/rehearsal_gpus$n.json').read().strip().splitlines()[-1]); print('gpus', d['n_gpus'], '%.3f M/s' % (d['value']/1e6), d['scaling'], d.get('rehearsal','')[:60])"
done
echo done
