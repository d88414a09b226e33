#!/bin/bash
cd $GRAFT_REPO_ROOT/profiles/micro
O=$GRAFT_REPO_ROOT/gpurun_out/micro; mkdir -p $O
{ GPU_MAX_HW_QUEUES=8 timeout -k 5 15 ./stream_handover 8 100 0; GPU_MAX_HW_QUEUES=8 timeout -k 5 15 ./stream_handover 4 150 3; GPU_MAX_HW_QUEUES=5 timeout -k 5 15 ./stream_handover 6 100 0; } > $O/stream_handover2.txt 2>&1
cat $O/stream_handover2.txt
