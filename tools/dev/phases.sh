#!/bin/bash
# dev only: phase stamps of the ICP kernel (build with -DFL_ICP_PHASES plus "$@", bench at B frames)
cd "$GRAFT_REPO_ROOT"
export FL_ICP_PHASES=1
for v in "$@"; do
  B=${B:-1280} ARGS="--templates 360 --no-extras ${ARGS:-}" bash tools/dev/variants.sh "-DFL_ICP_PHASES $v"
done
