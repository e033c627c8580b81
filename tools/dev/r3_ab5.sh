#!/bin/bash
# dev only (round 3): chain-wave priority under the pipelined search; phases of search 2 / 3 at 2048 and 2560 in one call
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES -DFL_ICP_SEARCH=2|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3 -DFL_ICP_CHAIN_PRIO=3|2048" "-DFL_ICP_PHASES -DFL_ICP_SEARCH=3 -DFL_ICP_CHAIN_PRIO=3|2560" 2>&1 | grep -o "^\[.*whole kernel [0-9.]*\|\"value.*" | tee gpurun_out/r3_ab5_phases.log
unset FL_ICP_PHASES
bash tools/dev/ab.sh "-DFL_ICP_SEARCH=3|2048" "-DFL_ICP_SEARCH=3 -DFL_ICP_CHAIN_PRIO=3|2048" "-DFL_ICP_SEARCH=3 -DFL_ICP_CHAIN_PRIO=1|2048" "-DFL_ICP_SEARCH=2 -DFL_ICP_CHAIN_PRIO=3|2048" "-DFL_ICP_SEARCH=3|2560" "-DFL_ICP_SEARCH=3 -DFL_ICP_CHAIN_PRIO=3|2560" 2>&1 | tee gpurun_out/r3_ab5.log
