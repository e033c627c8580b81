#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES|2048" "-DFL_ICP_PHASES|256" 2>&1 | grep -o "icp phase Mcyc.*whole kernel [0-9.]*\|\"value.*" | tee gpurun_out/r3_ab9_phases.log
