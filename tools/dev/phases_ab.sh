#!/bin/bash
# dev only: phase stamps for flag sets x batch on one box: phases_ab.sh "flags|batch" ...
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
for spec in "$@"; do
  flags="${spec%%|*}"; batch="${spec##*|}"
  B=$batch ARGS="--templates 360 --no-extras" bash tools/dev/variants.sh "-DFL_ICP_PHASES $flags"
done
