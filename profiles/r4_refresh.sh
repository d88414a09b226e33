#!/bin/bash
# round-4 evidence refresh for the build as it stands (after the last host-side change to the slot windows): GPU tests, the counter
# passes bench.py reads (hash-stamped), the kernel statistics of the default invocation, the default invocation itself
# -> gpurun_out/r4final/ (copied into profiles/r4_* by profiles/r4_fill.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final; mkdir -p $O
echo "== gpu tests"; timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
echo "== pmc rgb"; bash profiles/r4_pmc.sh r4final/pmc_rgb 4096 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_rgb.log 2>&1 || exit 1
echo "== stats rgb"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_rgb -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/kt_rgb.log 2>&1 ) || exit 1
f=$(find $O/kt_rgb -name '*.db' | head -1); echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras" > $O/kernel_stats_rgb.txt; python profiles/dbstats.py $f 36 >> $O/kernel_stats_rgb.txt
python profiles/dbtimeline.py $f k_commit 3 > $O/timeline_rgb.txt; rm -rf $O/kt_rgb
echo "== pmc perceptual"; bash profiles/r4_pmc.sh r4final/pmc_perceptual 2048 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --config perceptual > $O/pmc_perceptual.log 2>&1 || exit 1
echo "== pmc dither"; bash profiles/r4_pmc.sh r4final/pmc_dither 2048 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --config dither > $O/pmc_dither.log 2>&1 || exit 1
python profiles/r4_fill.py > /dev/null 2>&1   # bench.py reads profiles/r4_pmc_*.json: in place before the default invocation
echo "== bench"
t0=$(date +%s); python bench.py > $O/bench_rgb.json 2> $O/bench_rgb.err || exit 1; t1=$(date +%s); echo "default invocation: $((t1 - t0)) s" | tee $O/bench_rgb_wall.txt
tail -c 600 $O/bench_rgb.json
echo done
