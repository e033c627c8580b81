#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/dev/ab.sh "-DFL_ICP_SPEC=1|4096" "-DFL_ICP_SPEC=2|4096" "-DFL_ICP_SPEC=1|4096" "-DFL_ICP_SPEC=2|4096" "-DFL_ICP_SPEC=1|2560" "-DFL_ICP_SPEC=2|2560" 2>&1 | tee gpurun_out/r3_ab17.log
