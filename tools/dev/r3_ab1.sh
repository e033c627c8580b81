#!/bin/bash
# dev only (round 3): parity tests of the new organised search, then old / new A/B with phase stamps and histograms, one box
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_icp.py tests/test_gpu_golden_cadreco.py -x -q > gpurun_out/r3_ab1_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab1_tests.log
tail -3 gpurun_out/r3_ab1_tests.log
bash tools/dev/ab.sh "-DFL_ICP_SEARCH=1|2560" "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=1|2560" "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=1|8" "-DFL_ICP_SEARCH=2|8" 2>&1 | tee gpurun_out/r3_ab1.log
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=1|2560" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=2|2560" 2>&1 | tee gpurun_out/r3_ab1_phases.log
