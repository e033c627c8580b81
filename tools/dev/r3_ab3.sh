#!/bin/bash
# dev only (round 3): where a search step's time goes (segment stamps), with and without the deferred stores
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
bash tools/dev/ab.sh "-DFL_ICP_PHASES|2560" "-DFL_ICP_PHASES -DFL_ICP_LATE_STORE=1|2560" "-DFL_ICP_PHASES|256" "-DFL_ICP_PHASES|8" 2>&1 | tee gpurun_out/r3_ab3_phases.log
unset FL_ICP_PHASES
bash tools/dev/ab.sh "-DFL_ICP_LATE_STORE=0|2560" "-DFL_ICP_LATE_STORE=1|2560" "-DFL_ICP_LATE_STORE=0|2560" "-DFL_ICP_LATE_STORE=1|2560" 2>&1 | tee gpurun_out/r3_ab3.log
