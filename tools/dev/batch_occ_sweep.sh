#!/bin/bash
# dev only (round 3): ICP us per frame against batch size and workgroups per CU, longest-first order on
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --templates ${T:-360} --batch $1 2>&1 | grep -o "\"value[^,]*\|\"icp_ms[^,]*" | tr '\n' ' '; echo; }
for B in 2560 3840 4096 5120 6144; do
  for OCC in 4 5; do echo -n "[occ=$OCC b$B] "; FL_ICP_OCC=$OCC run $B; done
done 2>&1 | tee gpurun_out/r3_ab13.log
