#!/bin/bash
# dev only (tools/dev/README.md): build the library with different -D flags on the GPU box and bench each (usage: dev_variants.sh "flags1" "flags2" ...)
cd fealess_amd/csrc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
for v in "$@"; do
  rm -f *.o
  make -s CXXFLAGS="$BASE $v" 2>&1 | grep error
  (cd ../.. && timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --batch ${B:-1280} ${ARGS:-} 2>&1 | grep -o "icp phase.*\|icp organised.*\|\"value[^,]*\|stage_ms_last_step[^}]*" | sed "s/^/[$v] /" | cut -c1-400; echo)
done
