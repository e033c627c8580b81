#!/bin/bash
# dev only: A/B of build flag sets x batch sizes on ONE box (box-to-box spread is +-10 %): ab.sh "flags|batch" ...
cd "$GRAFT_REPO_ROOT/fealess_amd/csrc"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
for spec in "$@"; do
  flags="${spec%%|*}"; batch="${spec##*|}"
  rm -f fl_icp.o
  make -s CXXFLAGS="$BASE $flags" 2>&1 | grep error
  (cd ../.. && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --templates ${T:-360} --batch $batch ${ARGS:-} 2>&1 | grep -o "icp phase.*\|icp organised.*\|icp union.*\|icp search step.*\|icp workgroup timeline.*\|icp phase A2.*\|\"value[^,]*\|\"icp_ms[^,]*" | tr '\n' ' ' | sed "s/^/[$flags | $batch] /" | cut -c1-900; echo)
done
rm -f fl_icp.o; make -s 2>&1 | grep error
