#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sparse_path_equals_dense_path" > $O/a.log 2>&1; echo "default rc=$?"; tail -3 $O/a.log
SNES_BASE_STREAM=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sparse_path_equals_dense_path" > $O/b.log 2>&1; echo "nostream rc=$?"; tail -3 $O/b.log
