import os, sys, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O
from fealess_amd import synth
from test_abi_cpu import write_linemod_yaml, write_png16
import tempfile
tmp = tempfile.mkdtemp()
sc = synth.recognition_scene(lambda b, d, l: O.quantize_pyramid(b, d, l), levels=2, seed=13, n_views=3)
d = os.path.join(tmp, "obj"); os.makedirs(os.path.join(d, "depth"))
write_linemod_yaml(os.path.join(d, "linemod_templates.yml"), sc["bank"], [5, 8])
for i, md in enumerate(sc["bank"].model_depths):
    write_png16(os.path.join(d, "depth", f"{i}.png"), md)
cad = os.path.join(ROOT, "fealess_amd", "cadreco")
exe = os.path.join(tmp, "caller")
r = subprocess.run(["g++", "-std=c++14", "-O1", "-I", cad, os.path.join(ROOT, "tests", "dropin", "tu_caller_gpu.cpp"), "-o", exe,
                    "-L", cad, "-lcadreco_hip", "-Wl,-rpath," + cad, "-Wl,-rpath," + os.path.join(ROOT, "fealess_amd", "csrc")],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout)
bgr, depth = np.ascontiguousarray(sc["bgr"]), np.ascontiguousarray(sc["depth"])
bgr.tofile(os.path.join(tmp, "f.bgr")); depth.tofile(os.path.join(tmp, "f.d16"))
fx, fy, cx, cy = sc["K"]
for env_poison in ("1", None):
    env = dict(os.environ)
    if env_poison: env["FL_DEV_POISON"] = env_poison
    else: env.pop("FL_DEV_POISON", None)
    r = subprocess.run([exe, d, os.path.join(tmp, "f.bgr"), os.path.join(tmp, "f.d16"), "640", "480", repr(fx), repr(fy), repr(cx), repr(cy)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    print("poison", env_poison, r.returncode, r.stdout)
