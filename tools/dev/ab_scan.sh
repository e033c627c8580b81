#!/bin/bash
# dev only: A/B of build flag sets for fl_linemod.hip (scan / refine; OBJ=fl_frontend.o: the front-end) on ONE box: ab_scan.sh "flags" ...  (T, B, CONFIG from the environment)
cd "$GRAFT_REPO_ROOT/fealess_amd/csrc"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
for flags in "$@"; do
  rm -f ${OBJ:-fl_linemod.o}
  make -s CXXFLAGS="$BASE $flags" 2>&1 | grep error
  (cd ../.. && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --templates ${T:-2000} ${B:+--batch $B} ${CONFIG:+--config $CONFIG} 2>&1 | grep -o "\"value[^,]*\|\"scan_ms[^,]*\|\"refine_ms[^,]*\|\"frontend_ms[^,]*\|\"lazy_frontend_ms[^,]*" | tr '\n' ' ' | sed "s/^/[$flags] /" | cut -c1-300; echo)
done
rm -f ${OBJ:-fl_linemod.o}; make -s 2>&1 | grep error
