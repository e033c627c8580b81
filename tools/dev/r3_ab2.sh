#!/bin/bash
# dev only (round 3): old / new organised search A/B with phase stamps, then the whole GPU suite on the new build, one box
cd "$GRAFT_REPO_ROOT"
bash tools/dev/ab.sh "-DFL_ICP_SEARCH=1|2560" "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=1|2560" "-DFL_ICP_SEARCH=2|2560" "-DFL_ICP_SEARCH=1|8" "-DFL_ICP_SEARCH=2|8" "-DFL_ICP_SEARCH=1|2048" "-DFL_ICP_SEARCH=2|2048" 2>&1 | tee gpurun_out/r3_ab2.log
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=1|2560" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=2|2560" 2>&1 | tee gpurun_out/r3_ab2_phases.log
unset FL_ICP_PHASES
python -m pytest tests -m gpu -x -q > gpurun_out/r3_ab2_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab2_tests.log
tail -3 gpurun_out/r3_ab2_tests.log
