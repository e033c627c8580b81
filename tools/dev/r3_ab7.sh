#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=2|2048" 2>&1 | grep -o "icp phase A2.*barrier [0-9.]*\|icp phase Mcyc.*whole kernel [0-9.]*" | tee gpurun_out/r3_ab7_phases.log
