#!/bin/bash
# Dev aid: frames/s and ICP ms against extra per-frame workspace stride (FL_DEV_WS_PAD bytes).
for pad in 0 4096 12288 20480 36864 69632 180224 1; do
  FL_DEV_WS_PAD=$pad timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/sweep_$pad.log 2>&1 || exit 1
  python - "$pad" <<'PY'
import json, sys
pad = sys.argv[1]
j = json.loads(open(f"gpurun_out/sweep_{pad}.log").read().strip().splitlines()[-1])
s = j["stage_ms_last_step"]
print(pad, round(j["value"]), s["icp_ms"], s["frontend_ms"], s["scan_ms"], flush=True)
PY
done
