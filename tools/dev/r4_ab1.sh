#!/bin/bash
# dev only (round 4): GPU suite on the default build, then A/B of the exact block sums / chain-SIMD election on ONE box
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4_t1.log 2>&1; tail -3 gpurun_out/r4_t1.log
T=2000 bash tools/dev/ab.sh "-DFL_ICP_BSUM=0 -DFL_ICP_CHAIN_SIMD=0|4096" "-DFL_ICP_BSUM=1 -DFL_ICP_CHAIN_SIMD=0|4096" "-DFL_ICP_BSUM=0 -DFL_ICP_CHAIN_SIMD=1|4096" "-DFL_ICP_BSUM=1 -DFL_ICP_CHAIN_SIMD=1|4096" "-DFL_ICP_BSUM=0 -DFL_ICP_CHAIN_SIMD=0|4096" "-DFL_ICP_BSUM=1 -DFL_ICP_CHAIN_SIMD=1|4096" "-DFL_ICP_BSUM=0 -DFL_ICP_CHAIN_SIMD=0|8" "-DFL_ICP_BSUM=1 -DFL_ICP_CHAIN_SIMD=1|8" 2>&1 | tee gpurun_out/r4_ab1.txt
