#!/bin/bash
# dev only (round 3): depth-staged pipelined search: parity tests, then A/B against the previous library (search 2) on one box
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_golden_cadreco.py -x -q > gpurun_out/r3_ab8_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab8_tests.log
tail -3 gpurun_out/r3_ab8_tests.log
BATCHES="2048 2560 8" bash tools/dev/ab_lib.sh 2>&1 | tee gpurun_out/r3_ab8.log
