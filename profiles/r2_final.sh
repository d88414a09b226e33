#!/bin/bash
# round-2 evidence run: PMC passes and kernel-trace stats per configuration, bench lines, shard proxy -> gpurun_out/r2final/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2final; mkdir -p $O
if [ -z "$R2_BENCH_ONLY" ]; then
bash profiles/r2_pmc.sh r2final/pmc_rgb default 4096 > $O/pmc_rgb.log 2>&1
bash profiles/r2_pmc.sh r2final/pmc_perceptual default 2048 --config perceptual > $O/pmc_perceptual.log 2>&1
bash profiles/r2_pmc.sh r2final/pmc_dither default 2048 --config dither > $O/pmc_dither.log 2>&1
bash profiles/r2_pmc.sh r2final/pmc_images 1 2048 --config images > $O/pmc_images.log 2>&1
python profiles/r2_fill.py --pmc-only   # the bench lines below read traffic / VALU counts from profiles/r2_pmc_<config>.json
echo pmc done
stats() { # name, lanes, bench args
  n=$1; l=$2; shift 2
  ( cd /tmp && [ "$l" != default ] && export SNES_LANES=$l; cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$n -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$O/kt_$n.log 2>&1 )
  f=$(find $O/kt_$n -name '*.db' | head -1); python profiles/dbstats.py $f 30 > $O/kernel_stats_$n.txt
}
stats rgb default
stats rgb_lanes2 2
stats rgb_batch64 default --batch 64 --steps 200
stats perceptual default --config perceptual
stats dither default --config dither --steps 10 --warmup 2
stats images default --config images --steps 20 --warmup 3
echo stats done
fi
python bench.py > $O/bench_rgb.json 2> $O/bench_rgb.err
python bench.py --config perceptual --steps 100 > $O/bench_perceptual.json 2> $O/bench_perceptual.err
python bench.py --config dither --steps 40 > $O/bench_dither.json 2> $O/bench_dither.err
python bench.py --config images --steps 60 > $O/bench_images.json 2> $O/bench_images.err
python bench.py --config images --perceptual --steps 30 > $O/bench_images_perceptual.json 2> $O/bench_images_perceptual.err
python bench.py --config images --batch 256 --steps 30 > $O/bench_images_b256.json 2> $O/bench_images_b256.err
python bench.py --batch 1024 --steps 200 --no-cpu-baseline > $O/bench_rgb_b1024.json 2> $O/bench_rgb_b1024.err
python bench.py --batch 8192 --steps 100 --no-cpu-baseline > $O/bench_rgb_b8192.json 2> $O/bench_rgb_b8192.err
python profiles/shard_proxy.py > $O/shard_proxy_rgb.json 2> $O/shard_proxy_rgb.err
python profiles/shard_proxy.py --totals 16384,32768 --steps 40 > $O/shard_proxy_rgb_large.json 2> $O/shard_proxy_rgb_large.err
echo bench done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2final/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "ms/step %.3f"%d["ms_per_step"], "frac", round(d["roofline"]["frac"],4), "ref64", d.get("reference_batch",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
