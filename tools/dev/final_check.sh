#!/bin/bash
# dev only: the whole GPU suite + the default bench line, summary printed
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_all.log 2>&1; tail -3 gpurun_out/gpu_all.log
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["detections"], d["icp_points_mean"], d["stage_ms_last_step"])
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["distinct_frames"])
PY
