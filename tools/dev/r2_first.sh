#!/bin/bash
# dev only: GPU check of a changed ICP kernel: parity tests, then phase stamps at full batch, then bench at three batch sizes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_icp.py -x -q > gpurun_out/icp_tests.log 2>&1; rc=$?
tail -3 gpurun_out/icp_tests.log
[ $rc -ne 0 ] && exit $rc
for b in 1280 8 1; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --templates 360 --batch $b > gpurun_out/b$b.json 2> gpurun_out/b$b.err || { tail -5 gpurun_out/b$b.err; exit 1; }
  grep -o "\"value[^,]*\|stage_ms_last_step[^}]*" gpurun_out/b$b.json | tr '\n' ' '; echo
done
FL_ICP_PHASES=1 B=1280 ARGS="--templates 360 --no-extras" bash tools/dev/variants.sh "-DFL_ICP_PHASES" "$@"
