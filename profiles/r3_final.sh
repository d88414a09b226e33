#!/bin/bash
# round-3 evidence run -> gpurun_out/r3final/ (copied into profiles/r3_* by profiles/r3_fill.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3final; mkdir -p $O
step() { echo "== $1"; }
if [ -z "$R3_SKIP_PMC" ]; then
step "pmc rgb"; bash profiles/r3_pmc.sh r3final/pmc_rgb 4096 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_rgb.log 2>&1 || exit 1
cp $O/pmc_rgb/pmc.json profiles/r3_pmc_rgb.json
step "pmc perceptual"; bash profiles/r3_pmc.sh r3final/pmc_perceptual 2048 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --config perceptual > $O/pmc_perceptual.log 2>&1 || exit 1
cp $O/pmc_perceptual/pmc.json profiles/r3_pmc_perceptual.json
step "pmc dither"; bash profiles/r3_pmc.sh r3final/pmc_dither 2048 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --config dither > $O/pmc_dither.log 2>&1 || exit 1
cp $O/pmc_dither/pmc.json profiles/r3_pmc_dither.json
# (the slot windows under --pmc: rocprofv3's counter tool crashes — SIGSEGV inside the tool at the first launch after a window has used its side stream — so the windows are profiled by kernel trace only)
fi
stats() { # name, script, args
  n=$1; sc=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$n -o p -- python3 $GRAFT_REPO_ROOT/$sc "$@" > $GRAFT_REPO_ROOT/$O/kt_$n.log 2>&1 ) || return 1
  f=$(find $O/kt_$n -name '*.db' | head -1); echo "# rocprofv3 --kernel-trace --stats -- python3 $sc $*" > $O/kernel_stats_$n.txt; python profiles/dbstats.py $f 34 >> $O/kernel_stats_$n.txt
  [ -n "$TL" ] && python profiles/dbtimeline.py $f $TL 3 > $O/timeline_$n.txt
  rm -rf $O/kt_$n
}
step "stats"
stats rgb bench.py --no-cpu-baseline --no-extras || exit 1   # the default invocation's timed region (400 steps, 10 warm-up): the V pass's average must agree with roofline.avg_launch_ms
stats perceptual bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras --config perceptual || exit 1
TL=k_commit stats dither bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --config dither || exit 1   # (timeline: a call three from the end)
bash profiles/r3_steps.sh r3final/steps_dither k_commit "k_dither4ILi15ELi1 k_ditherw k_sparse_v2E k_sparse_h2E k_sparse_down1 k_dither_diff" bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --config dither > /dev/null || exit 1
cp $O/steps_dither/steps.txt $O/steps_dither.txt
python profiles/r3_acceptance.py > $O/acceptance_rgb.txt 2>/dev/null || exit 1
stats images bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --config images || exit 1
TL=kw_commit stats slots profiles/r3_slots.py --converge 100 --calls 960 --window 64 || exit 1
TL=kw_commit stats slots_dither profiles/r3_slots.py --converge 100 --calls 480 --window 64 --config dither || exit 1
step "bench"
t0=$(date +%s); python bench.py > $O/bench_rgb.json 2> $O/bench_rgb.err || exit 1; t1=$(date +%s); echo "default invocation: $((t1 - t0)) s" | tee $O/bench_rgb_wall.txt
python bench.py --config perceptual --steps 100 --no-config-extras > $O/bench_perceptual.json 2> $O/bench_perceptual.err || exit 1
python bench.py --config dither --steps 40 --no-config-extras > $O/bench_dither.json 2> $O/bench_dither.err || exit 1
python bench.py --config images --steps 60 > $O/bench_images.json 2> $O/bench_images.err || exit 1
SNES_BENCH_FORCE_DIST=1 python bench.py --no-config-extras --no-cpu-baseline --steps 100 > $O/bench_rgb_rccl_one_rank.json 2> $O/bench_rgb_rccl_one_rank.err || exit 1
step "slots"
python profiles/r3_slots.py --converge 30 --calls 1920 > $O/slots_rgb_c30.json 2>/dev/null || exit 1
python profiles/r3_slots.py --converge 100 --calls 1920 > $O/slots_rgb_c100.json 2>/dev/null || exit 1
python profiles/r3_slots.py --converge 100 --calls 1920 --window 1 > $O/slots_rgb_c100_call_by_call.json 2>/dev/null || exit 1
python profiles/r3_slots.py --converge 60 --calls 960 --config perceptual > $O/slots_perceptual_c60.json 2>/dev/null || exit 1
python profiles/r3_slots.py --converge 60 --calls 960 --config dither > $O/slots_dither_c60.json 2>/dev/null || exit 1
python profiles/r3_dither_perceptual.py 2048 16 > $O/dither_perceptual.json 2>/dev/null || exit 1
python profiles/r3_slots.py --config dither_perceptual --converge 1 --calls 240 > $O/slots_dither_perceptual_c1.json 2>/dev/null || exit 1
python profiles/r3_slots.py --config dither_perceptual --converge 1 --calls 120 --window 1 > $O/slots_dither_perceptual_c1_call_by_call.json 2>/dev/null || exit 1
step "proxy"
python profiles/shard_proxy.py --totals 64,4096,32768 --steps 40 --windows 64,128,256,480 > $O/shard_proxy.json 2> $O/shard_proxy.err || exit 1
echo done
