#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_groups" > $O/pytest_lg.log 2>&1; echo "rc=$?"; tail -25 $O/pytest_lg.log
