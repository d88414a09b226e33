#!/bin/bash
# round-4 evidence run, part $1 (1 | 2 | 3) -> gpurun_out/r4final/ (copied into profiles/r4_* by profiles/r4_fill.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final; mkdir -p $O
step() { echo "== $1"; }
stats() { # name, script, args
  n=$1; sc=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$n -o p -- python3 $GRAFT_REPO_ROOT/$sc "$@" > $GRAFT_REPO_ROOT/$O/kt_$n.log 2>&1 ) || return 1
  f=$(find $O/kt_$n -name '*.db' | head -1); echo "# rocprofv3 --kernel-trace --stats -- python3 $sc $*" > $O/kernel_stats_$n.txt; python profiles/dbstats.py $f 36 >> $O/kernel_stats_$n.txt
  [ -n "$TL" ] && python profiles/dbtimeline.py $f $TL 3 > $O/timeline_$n.txt
  rm -rf $O/kt_$n
}
if [ "$1" = "1" ]; then
step "pmc rgb"; bash profiles/r4_pmc.sh r4final/pmc_rgb 4096 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_rgb.log 2>&1 || exit 1
step "pmc perceptual"; bash profiles/r4_pmc.sh r4final/pmc_perceptual 2048 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --config perceptual > $O/pmc_perceptual.log 2>&1 || exit 1
step "pmc dither"; bash profiles/r4_pmc.sh r4final/pmc_dither 2048 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --config dither > $O/pmc_dither.log 2>&1 || exit 1
# the slot windows under the counter tool: the host waits for the device every 60 calls of the converge phase (without that the
# tool dies behind ~20 k unsynchronised launches: profiles/micro/pmc_queue_repro.hip shows it without the library)
step "pmc slots"; bash profiles/r4_pmc.sh r4final/pmc_slots 64 profiles/r4_slots.py --converge 30 --calls 480 --window 64 --sync-every 60 > $O/pmc_slots.log 2>&1 || exit 1
step "sq"; bash profiles/r4_pmc_sq.sh r4final/sq4096 4096 bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline > $O/sq4096.log 2>&1
bash profiles/r4_pmc_sq.sh r4final/sq64 64 bench.py --batch 64 --steps 40 --warmup 8 --no-extras --no-cpu-baseline > $O/sq64.log 2>&1
fi
if [ "$1" = "2" ]; then
step "stats"
TL=k_commit stats rgb bench.py --no-cpu-baseline --no-extras || exit 1   # the default invocation's timed region (400 steps, 10 warm-up): the V pass's average must agree with roofline.avg_launch_ms
TL=k_commit stats rgb_batch64 bench.py --batch 64 --steps 200 --no-cpu-baseline --no-extras || exit 1
stats perceptual bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-extras --config perceptual || exit 1
TL=k_commit stats dither bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --config dither || exit 1
stats images bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --config images || exit 1
TL=kw_commit stats slots profiles/r4_slots.py --converge 100 --calls 960 --window 64 || exit 1
TL=kw_commit stats slots_start profiles/r4_slots.py --converge 0 --calls 960 || exit 1
TL=kw_commit stats slots_dither profiles/r4_slots.py --converge 100 --calls 480 --window 64 --config dither || exit 1
step "bench"
t0=$(date +%s); python bench.py > $O/bench_rgb.json 2> $O/bench_rgb.err || exit 1; t1=$(date +%s); echo "default invocation: $((t1 - t0)) s" | tee $O/bench_rgb_wall.txt
fi
if [ "$1" = "3" ]; then
step "bench others"
python bench.py --config perceptual --steps 100 --no-config-extras > $O/bench_perceptual.json 2> $O/bench_perceptual.err || exit 1
python bench.py --config dither --steps 40 --no-config-extras > $O/bench_dither.json 2> $O/bench_dither.err || exit 1
python bench.py --config images --steps 60 > $O/bench_images.json 2> $O/bench_images.err || exit 1
SNES_BENCH_FORCE_DIST=1 python bench.py --no-config-extras --no-cpu-baseline --steps 100 > $O/bench_rgb_rccl_one_rank.json 2> $O/bench_rgb_rccl_one_rank.err || exit 1
python bench.py --batch 64 --steps 400 --no-extras --no-cpu-baseline > $O/bench_rgb_batch64.json 2> $O/bench_rgb_batch64.err || exit 1
step "slots"
python profiles/r4_slots.py --converge 30 --calls 1920 > $O/slots_rgb_c30.json 2>/dev/null || exit 1
python profiles/r4_slots.py --converge 100 --calls 1920 > $O/slots_rgb_c100.json 2>/dev/null || exit 1
python profiles/r4_slots.py --converge 100 --calls 1920 --window 1 > $O/slots_rgb_c100_call_by_call.json 2>/dev/null || exit 1
python profiles/r4_slots.py --converge 60 --calls 960 --config perceptual > $O/slots_perceptual_c60.json 2>/dev/null || exit 1
python profiles/r4_slots.py --converge 60 --calls 960 --config dither > $O/slots_dither_c60.json 2>/dev/null || exit 1
step "proxy"
python profiles/shard_proxy.py --totals 64,4096,32768 --steps 40 --windows 64,128,256,480 > $O/shard_proxy_rgb.json 2> $O/shard_proxy.err || exit 1
python profiles/shard_proxy.py --config dither --totals 64,4096 --steps 20 --windows 64,256 --converge 10 > $O/shard_proxy_dither.json 2>> $O/shard_proxy.err || exit 1
python profiles/shard_proxy.py --config perceptual --totals 64,4096 --steps 30 --windows 64,256 --converge 10 > $O/shard_proxy_perceptual.json 2>> $O/shard_proxy.err || exit 1
python profiles/r4_acceptance.py > $O/acceptance_rgb.txt 2>/dev/null
fi
echo done
