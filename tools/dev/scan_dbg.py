"""dev: raw similarity maps of the cluttered single-level case under the library FEALESS_HIP_LIB points to -> gpurun_out/scan_maps_<tag>.npy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from fealess_amd import api, synth
tag = sys.argv[1]
rng = np.random.default_rng(4)
qs = [synth.random_quantized(rng, 320, 240, 0.9)]
bank = synth.make_bank("obj", 40, 1, 1, 320, 240, seed=9, bbox=64)
ctx = api.Context(0)
det = api.Detector(ctx, 1, [8]); det.add_class(bank); det.finalize(320, 240, max_batch=1)
got, n = det.match_quantized(qs, 50.0)
maps = det.similarity_maps(0, 40)
np.save(os.path.join(ROOT, "gpurun_out", f"scan_maps_{tag}.npy"), maps)
print(tag, n, maps.sum())
if tag == "prev":
    a = np.load(os.path.join(ROOT, "gpurun_out", "scan_maps_new.npy"))
    d = np.argwhere(a != maps)
    print("differing cells:", len(d))
    for t, y, x in d[:40]:
        print("template", t, "cell", y * 40 + x, "new", a[t, y, x], "prev", maps[t, y, x])
