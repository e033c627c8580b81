#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES|4096" "-DFL_ICP_PHASES|2560" 2>&1 | grep -o "^\[[^]]*\]\|icp phase Mcyc.*whole kernel [0-9.]*\|icp search step segments[^i]*\|icp workgroup timeline.*span: [0-9]*\|\"value.*" | tee gpurun_out/r3_ab16_phases.log
