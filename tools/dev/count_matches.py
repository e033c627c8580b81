"""dev: coarse candidates and final matches per frame of the bench scenes (2000 templates), via fl_frame_counters.
usage (GPU box): python tools/dev/count_matches.py [templates] [frames]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import bench
from fealess_amd import api

sys.argv = [sys.argv[0]] + ["--templates", sys.argv[1] if len(sys.argv) > 1 else "2000", "--batch", sys.argv[2] if len(sys.argv) > 2 else "64",
                            "--no-extras", "--no-cpu-baseline"]
args = bench.parse()
ctx = api.Context(0)
w, h, K = bench.geometry(args)
bank, scenes = bench.build_bank(ctx, args, args.templates, w, h, K)
bgrs, depths = bench.build_frames(scenes, args.batch, 0, w, h)
run = bench.Runner(ctx, args, bank, bgrs, depths, w, h, K)
run.timed(1, 0)
res, _ = run.collect()
cands, matches = [], []
for f in range(args.batch):
    cnt = (C.c_int32 * 4)()
    run.det.lib.fl_frame_counters(run.det.h, f, cnt)
    cands.append(cnt[0]); matches.append(cnt[1])
cands, matches = np.array(cands), np.array(matches)
print("coarse candidates per frame: mean %.0f (min %d, max %d); matches after refinement + sort/unique: mean %.1f (min %d, max %d)" %
      (cands.mean(), cands.min(), cands.max(), matches.mean(), matches.min(), matches.max()))
run.close(); ctx.close()
