#!/bin/bash
# dev only: time ICP phase A1 under ablations (built on the GPU box)
cd fealess_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
for v in "-DFL_ICP_DEBUG" "-DFL_NONE"; do
  rm -f fl_icp.o
  make -s CXXFLAGS="$BASE $v" 2>&1 | grep error
  for b in 64 1024; do
  (cd ../.. && timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b 2>&1 | grep -o "icp dbg.*\|icp_ms[^,]*" | tail -2 | sed "s/^/[$v $b] /" | cut -c1-220)
  done
done
