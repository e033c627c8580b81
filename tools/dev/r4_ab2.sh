#!/bin/bash
# dev only (round 4): block-sum variants in the 256-thread kernel and the occupancy experiment (3 / 2 / 1 workgroups per CU by LDS padding)
cd "$GRAFT_REPO_ROOT"
T=2000 bash tools/dev/ab.sh "|4096" "-DFL_ICP_BSUM=2 -DFL_ICP_ALLPROD=0|4096" "-DFL_ICP_LDS_PAD=12000|4096" "-DFL_ICP_LDS_PAD=26000|4096" "-DFL_ICP_LDS_PAD=60000|4096" "|4096" "-DFL_ICP_BSUM=2 -DFL_ICP_ALLPROD=0|4096" "-DFL_ICP_LDS_PAD=12000|3072" "|3072" 2>&1 | tee gpurun_out/r4_ab2.txt
