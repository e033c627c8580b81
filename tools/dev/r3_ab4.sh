#!/bin/bash
# dev only (round 3): the pipelined organised search (FL_ICP_SEARCH=3): parity tests, then A/B against FL_ICP_SEARCH=2, one box
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_icp.py tests/test_gpu_golden_cadreco.py -x -q > gpurun_out/r3_ab4_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab4_tests.log
tail -3 gpurun_out/r3_ab4_tests.log
bash tools/dev/ab.sh "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=3|2560" "-DFL_ICP_SEARCH=2|2048" "-DFL_ICP_SEARCH=3|2048" "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=3|2560" 2>&1 | tee gpurun_out/r3_ab4.log
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2560" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" 2>&1 | tee gpurun_out/r3_ab4_phases.log
