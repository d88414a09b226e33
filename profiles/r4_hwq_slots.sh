#!/bin/bash
# GPU_MAX_HW_QUEUES against the reference's loop through the slot windows (adaptive windows, from the k-means start and converged) and call by call
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist_gap; mkdir -p $O
slots() { python profiles/r4_slots.py --converge 30 --calls 960 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['from_kmeans_start']; b=d['converged']
print('$TAG slots $* start %6.0f calls/s %.3f M useful | converged %6.0f calls/s %.3f M useful' % (a['calls_per_s'], a['useful_cand_per_s']/1e6, b['calls_per_s'], b['useful_cand_per_s']/1e6))" | tee -a $O/log4.txt; }
for i in 1 2; do
for q in 2 3 4 5 6 8; do TAG="hwq$q" GPU_MAX_HW_QUEUES=$q slots; done
done
for q in 2 4 8; do TAG="hwq$q" GPU_MAX_HW_QUEUES=$q slots --window 1; done
for q in 2 4 8; do TAG="hwq$q" GPU_MAX_HW_QUEUES=$q slots --config perceptual --calls 480; done
