#!/bin/bash
# dev only: ICP launch time of the 256- and the 1024-thread kernel over the batch sizes in between (FL_ICP_WIDE forces the width)
cd "$GRAFT_REPO_ROOT"
for B in ${BATCHES:-256 384 512 768 1024}; do
for wide in 0 1; do
  FL_ICP_WIDE=$wide timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --templates 360 --batch $B 2>&1 | grep -o "\"value[^,]*\|\"icp_ms[^,]*" | tr '\n' ' ' | sed "s|^|[b$B wide=$wide] |"; echo
done; done
