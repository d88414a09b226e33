#!/bin/bash
cd $GRAFT_REPO_ROOT/profiles/micro
O=$GRAFT_REPO_ROOT/gpurun_out/micro; mkdir -p $O
{ timeout -k 5 12 ./stream_handover 6 150 0; GPU_MAX_HW_QUEUES=4 timeout -k 5 12 ./stream_handover 6 150 1; GPU_MAX_HW_QUEUES=8 timeout -k 5 12 ./stream_handover 6 150 0; GPU_MAX_HW_QUEUES=2 timeout -k 5 12 ./stream_handover 6 150 0; } > $O/stream_handover.txt 2>&1
cat $O/stream_handover.txt
