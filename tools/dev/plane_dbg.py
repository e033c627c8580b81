import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import test_gpu_icp as T
import p2plane_model as P
from fealess_amd import api, _lib as L
ctx = api.Context(0)
for seed in range(3):
    c = T._plane_case(seed, *T.PLANE_CASES[seed])
    K = (608.0, 608.0, 320.0, 240.0)
    args = (c["model"], c["scene"], K, c["rect_model"], c["rect_ref"], 20, 0.0, -3.0e38, c["Rm"], c["tm"])
    got = ctx.detection(*args, L.FL_ICP_POINT_TO_PLANE)
    exp = P.detection_point_to_plane(*args)
    print(seed, "angle", T._angle_deg(got["R_final"], exp["R_final"]), "dT", np.abs(got["T_final"] - exp["T_final"]).max(),
          "ncorr", got["icp"]["n_corr_last"], exp["icp"]["n_corr_last"], "dm", got["icp"]["dist_mean"], exp["icp"]["dist_mean"])
