#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2560" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|1024" 2>&1 | grep -o "^\[[^]]*\]\|icp phase Mcyc.*whole kernel [0-9.]*\|icp workgroup timeline.*span: [0-9]*\|\"value.*" | tee gpurun_out/r3_ab11_phases.log
