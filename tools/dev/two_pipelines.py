#!/usr/bin/env python3
"""dev: TWO pipelines on one GPU -- two contexts (a HIP stream each) with a detector each, half of the frames each, their
steps submitted alternately, so that one pipeline's LINEMOD stages run beside the other's ICP launch.  The ICP launch normally
fills every CU's registers (4 workgroups x 128 VGPRs x 4 waves); option icp_wg_per_cu caps it so that the other stream's
kernels find room.  usage: two_pipelines.py <frames per pipeline> <icp_wg_per_cu, 0 = uncapped> [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
from fealess_amd import api

B = int(sys.argv[1]); cap = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sys.argv = [sys.argv[0], "--batch", str(B), "--templates", "2000"]
args = bench.parse()
w, h, K = bench.geometry(args)
ctx0 = api.Context(0)
bank, scenes = bench.build_bank(ctx0, args, args.templates, w, h, K)
bgrs, depths = bench.build_frames(scenes, 2 * B, 0, w, h)
ctxs = [ctx0, api.Context(0)]
runs = []
for k, c in enumerate(ctxs):
    c.set_option("icp_wg_per_cu", cap)
    runs.append(bench.Runner(c, args, bank, bgrs[k * B:(k + 1) * B], depths[k * B:(k + 1) * B], w, h, K))
def sync():
    for r in runs: r.sync()
for r in runs: r.step()
sync()
# one pipeline alone (the usual bench step, B frames)
t0 = time.perf_counter()
for _ in range(steps): runs[0].step()
sync(); one = time.perf_counter() - t0
# both, alternating
t0 = time.perf_counter()
for _ in range(steps):
    runs[0].step(); runs[1].step()
sync(); both = time.perf_counter() - t0
res, t = runs[0].collect()
print(f"frames per pipeline {B}, icp_wg_per_cu {cap}: one pipeline {B * steps / one:.0f} frames/s ({one / steps * 1e3:.2f} ms/step, ICP {t['icp_ms']:.2f} ms); "
      f"two pipelines {2 * B * steps / both:.0f} frames/s ({both / steps * 1e3:.2f} ms per pair of steps); found {sum(int(r.found) for r in res)}/{B}", flush=True)
