#!/bin/bash
# the --dither / --perceptual-palettes lines of part 3 again on the last build (their figures on the build before it were the slow draw of the queue lottery)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
timeout -k 5 60 python bench.py --config dither --steps 40 --no-config-extras > $O/bench_dither.json 2> $O/bench_dither.err || exit 1
timeout -k 5 60 python bench.py --config perceptual --steps 100 --no-config-extras > $O/bench_perceptual.json 2> $O/bench_perceptual.err || exit 1
timeout -k 5 40 python profiles/r4_slots.py --converge 60 --calls 960 --config dither > $O/slots_dither_c60.json 2>/dev/null || exit 1
timeout -k 5 40 python profiles/r4_slots.py --converge 60 --calls 960 --config perceptual > $O/slots_perceptual_c60.json 2>/dev/null || exit 1
echo done
