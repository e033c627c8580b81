#!/bin/bash
# dev only (round 3): longest-first job order of the ICP launch (FL_ICP_ORDER=0/1, same library), workgroup timeline, parity tests
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --templates ${T:-360} --batch $1 2>&1 | grep -o "icp workgroup timeline.*span: [0-9]*\|\"value[^,]*\|\"icp_ms[^,]*" | tr '\n' ' '; echo; }
for B in 2048 2560 3072 4096; do
  for O in 0 1 0 1; do echo -n "[order=$O b$B] "; FL_ICP_ORDER=$O run $B; done
done 2>&1 | tee gpurun_out/r3_ab12.log
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_golden_cadreco.py -x -q > gpurun_out/r3_ab12_tests.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_ab12_tests.log
tail -3 gpurun_out/r3_ab12_tests.log
