"""Dev aid: wall time of fl_extract_template_pyramid (Detector::addTemplate) per training view, GPU vs the CPU oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (runtime load order, see tests/conftest.py)
from fealess_amd import api, synth
import oracle_py as O

ctx = api.Context(0)
views = []
rng = np.random.default_rng(5)
for s in range(12):
    R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)), tz=float(rng.uniform(520, 700)),
                             yaw=float(rng.uniform(-1, 1)), tilt=float(rng.uniform(0.1, 0.6)))
    d, b, m = synth.render(640, 480, R, t, seed=s, noise=False)
    views.append((b, d, (m * 255).astype(np.uint8)))
for name, fn in (("gpu", lambda v: ctx.extract_template_pyramid(v[0], v[1], v[2], 2)), ("oracle", lambda v: O.add_template(v[0], v[1], v[2], 2))):
    fn(views[0])
    t0 = time.perf_counter()
    outs = [fn(v) for v in views]
    dt = (time.perf_counter() - t0) / len(views)
    print(f"{name}: {dt * 1e3:.2f} ms per view", flush=True)
