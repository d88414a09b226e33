#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2t; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log
