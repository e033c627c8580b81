#!/bin/bash
# dev only (round 3): what do the staged cache lines cost?  every transfer issued twice (the extra one over other lines of the frame)
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3 -DFL_ICP_HACK_DOUBLE=1|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3 -DFL_ICP_HACK_DOUBLE=1|2048" 2>&1 | grep -o "^\[.*whole kernel [0-9.]*\|\"value.*" | tee gpurun_out/r3_ab6_phases.log
