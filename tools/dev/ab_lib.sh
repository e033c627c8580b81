#!/bin/bash
# dev only: A/B of two builds of the library on one box: the in-tree one and fealess_amd/csrc/libfealess_hip_prev.so
cd "$GRAFT_REPO_ROOT"
for B in ${BATCHES:-2048}; do
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/fealess_amd/csrc/libfealess_hip_prev.so"; do
  FEALESS_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --templates ${T:-360} --batch $B ${ARGS:-} 2>&1 | grep -o "\"value[^,]*\|\"icp_ms[^,]*\|\"scan_ms[^,]*\|\"linmem_ms[^,]*\|\"frontend_ms[^,]*\|\"lazy_frontend_ms[^,}]*" | tr '\n' ' ' | sed "s|^|[${lib:+prev}${lib:-new } b$B] |" | sed "s|prev/root[^ ]*|prev|"; echo
done; done; done
