#!/bin/bash
# dev only (round 3): rotation-free producer loops of the chain phases: parity tests, then A/B against the previous library
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_golden_cadreco.py -x -q > gpurun_out/r3_ab10_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab10_tests.log
tail -3 gpurun_out/r3_ab10_tests.log
BATCHES="2048 2560 8" bash tools/dev/ab_lib.sh 2>&1 | tee gpurun_out/r3_ab10.log
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=2|2048" 2>&1 | grep -o "^\[[^]]*\]\|icp phase Mcyc.*whole kernel [0-9.]*\|\"value.*" | tee gpurun_out/r3_ab10_phases.log
