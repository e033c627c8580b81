#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/dev/ab.sh "-DFL_ICP_A2_DEEP=0|4096" "-DFL_ICP_A2_DEEP=1|4096" "-DFL_ICP_A2_DEEP=0|4096" "-DFL_ICP_A2_DEEP=1|4096" "-DFL_ICP_A2_DEEP=0|2560" "-DFL_ICP_A2_DEEP=1|2560" 2>&1 | tee gpurun_out/r3_ab18.log
