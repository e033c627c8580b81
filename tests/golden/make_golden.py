#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the ORACLE (oracle/liboracle.so).

The reference ships no golden vectors, known-answer tests or fixtures for this path and cannot
be built or run here (OpenCV 3.x is absent), so these vectors pin the oracle *restatement*
against regressions and give the HIP path a data-only target that travels to the GPU box --
they do NOT pin the oracle to the reference ("parity unpinned").
Run:  python tests/golden/make_golden.py     (inputs are seeded; output is deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_py as O  # noqa: E402
from fealess_amd import synth  # noqa: E402


def bank_arrays(bank):
    t, f, p = bank.arrays()
    return dict(templates=t, features=f, poses=p)


def linemod_fixture():
    rng = np.random.default_rng(20261003)
    w0, h0, T = 320, 160, [5, 8]
    qs = [synth.random_quantized(rng, w0 >> l, h0 >> l, 0.04) for l in range(2) for _ in range(2)]
    bank = synth.make_bank("obj", 24, 2, 2, w0, h0, seed=5, qs=qs, planted_frac=0.3, bbox=64)
    matches, n = O.match_quantized(qs, w0, h0, T, [bank], 60.0)
    lms = [O.build_linear_memories(qs[2 + m], 8) for m in range(2)]
    sims = np.stack([O.total_similarity(lms, bank, g, 160, 80, 8) for g in range(bank.n_pyramids)])
    lm0 = O.build_linear_memories(qs[0], 5)
    np.savez_compressed(os.path.join(HERE, "linemod_320x160.npz"), w0=w0, h0=h0, T=np.array(T), q0=qs[0], q1=qs[1], q2=qs[2],
                        q3=qs[3], matches=matches, n_matches=n, sims=sims, lm_level0_mod0_crc=np.array([int(lm0.astype(np.uint64).sum()),
                                                                                                       int((lm0.astype(np.uint64) * (np.arange(lm0.size, dtype=np.uint64).reshape(lm0.shape) % 251)).sum())]),
                        threshold=60.0, **bank_arrays(bank))
    print("linemod fixture:", n, "matches")


def frontend_fixture():
    R, t = synth.object_pose(tx=-15, ty=8, tz=640, yaw=0.25)
    depth, bgr, _ = synth.render(640, 480, R, t, seed=11)
    d = np.ascontiguousarray(depth[150:342, 200:456])       # 256 x 192 window around the object
    b = np.ascontiguousarray(bgr[150:342, 200:456])
    qo = O.quantized_orientations(b, 10.0)
    qn = O.quantized_normals(d)
    pd = O.pyrdown_bgr(b)
    qo1 = O.quantized_orientations(pd, 10.0)
    np.savez_compressed(os.path.join(HERE, "frontend_256x192.npz"), bgr=b, depth=d, qo=qo, qn=qn, pyrdown=pd, qo1=qo1)
    print("frontend fixture: non-zero", int((qo != 0).sum()), int((qn != 0).sum()))


def icp_fixture():
    rng = np.random.default_rng(7)
    R, t = synth.object_pose(tz=650.0)
    depth, _, mask = synth.render(640, 480, R, t, seed=3, noise=True, background=False)
    ys, xs = np.nonzero(mask)
    sel = np.sort(rng.choice(len(ys), size=1500, replace=False))
    z = depth[ys[sel], xs[sel]].astype(np.float32)
    ref = np.stack([(xs[sel] - 320.0) / 608.0 * z, (ys[sel] - 240.0) / 608.0 * z, z], 1).astype(np.float32)
    dR = synth.rot_z(0.02) @ synth.rot_x(-0.015) @ synth.rot_y(0.01)
    c = ref.mean(0)
    model = ((ref - c) @ dR.T + c + np.array([1.5, -2.0, 1.0])).astype(np.float32)
    model += rng.normal(0, 0.3, model.shape).astype(np.float32)
    r32 = O.icp(ref, model, 12, 0.0, -3.0e38, accum64=False, trace=True)
    r64 = O.icp(ref, model, 12, 0.0, -3.0e38, accum64=True)
    rdef = O.icp(ref, model, 10, 0.5, 0.01)
    np.savez_compressed(os.path.join(HERE, "icp_1500.npz"), ref=ref, model=model, R32=r32["R"], T32=r32["T"], dm32=r32["dist_mean"],
                        trace32=r32["trace"], R64=r64["R"], T64=r64["T"], Rdef=rdef["R"], Tdef=rdef["T"], iters_def=rdef["iters"])
    print("icp fixture: iters", r32["iters"], "default-threshold iters", rdef["iters"])


def recognition_fixture():
    sc = synth.recognition_scene(lambda b, d, l: O.quantize_pyramid(b, d, l), levels=2, seed=21, n_views=3, n_random=5)
    res = O.recognition(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], 75.0, 10, 0.5, 0.01)
    res20 = O.recognition(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], 75.0, 20, -1.0, -3.0e38)
    md = np.stack(sc["bank"].model_depths)
    np.savez_compressed(os.path.join(HERE, "recognition_vga.npz"), bgr=sc["bgr"], depth=sc["depth"], K=np.array(sc["K"]),
                        model_depths=md, pose=res["pose"], best=np.array([res["best"]["x"], res["best"]["y"], res["best"]["template_id"]]),
                        best_sim=res["best"]["similarity"], n_matches=res["n_matches"], n_points=res["det"]["n_points"],
                        iters=res["det"]["icp"]["iters"], pose20=res20["pose"], **bank_arrays(sc["bank"]))
    print("recognition fixture: found", res["found"], "tid", res["best"]["template_id"], "iters", res["det"]["icp"]["iters"])


if __name__ == "__main__":
    linemod_fixture()
    frontend_fixture()
    icp_fixture()
    recognition_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
