#!/bin/bash
cd "$GRAFT_REPO_ROOT"
BATCHES="4096 2560 8" bash tools/dev/ab_lib.sh 2>&1 | tee gpurun_out/r3_ab15.log
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py -x -q > gpurun_out/r3_ab15_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab15_tests.log
tail -3 gpurun_out/r3_ab15_tests.log
