"""dev: randomized GPU-vs-oracle differential run (longer than the test-suite budget).
usage (GPU box): python tools/dev/fuzz_parity.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import oracle_py as O
from fealess_amd import api, synth, _lib as L

os.environ.setdefault("FL_DEV_POISON", "1")   # lazy fine levels: poison what the tile marking leaves out
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
ctx = api.Context(0)
t0 = time.time()
stats = dict(icp=0, frontend=0, recognition=0, linemod=0, extract=0, detection=0, fail=0)
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
seed = int(os.environ.get("FUZZ_SEED", "1000"))
last_print = t0
while time.time() - t0 < budget:
    seed += 1
    if time.time() - last_print > 45:                        # gpurun kills a command that is silent for 7 minutes
        last_print = time.time()
        print("progress:", stats, flush=True)
    rng = np.random.default_rng(seed)
    kind = seed % 6 if len(sys.argv) < 3 else int(sys.argv[2])
    try:
        if kind == 0:      # ICP on random paired clouds of random size / misalignment / noise / invalid points
            n = int(rng.integers(3, 9000))
            R, t = synth.object_pose(tz=float(rng.uniform(500, 800)))
            depth, _, mask = synth.render(640, 480, R, t, seed=seed, noise=True, background=False)
            ys, xs = np.nonzero(mask)
            sel = np.sort(rng.choice(len(ys), size=min(n, len(ys)), replace=False))
            z = depth[ys[sel], xs[sel]].astype(np.float32)
            ref = np.stack([(xs[sel] - 320.0) / 608.0 * z, (ys[sel] - 240.0) / 608.0 * z, z], 1).astype(np.float32)
            dR = synth.rot_z(rng.uniform(-.05, .05)) @ synth.rot_x(rng.uniform(-.04, .04)) @ synth.rot_y(rng.uniform(-.04, .04))
            c = ref.mean(0)
            model = ((ref - c) @ dR.T + c + rng.uniform(-4, 4, 3)).astype(np.float32)
            model += rng.normal(0, rng.uniform(0, 0.5), model.shape).astype(np.float32)
            m = int(rng.integers(max(3, len(model) // 2), len(model) + 1)) if len(model) > 6 else len(model)
            model = model[:m]
            for k in rng.integers(0, len(model), size=int(rng.integers(0, 4))):
                model[k] = (1.0, 2.0, 950.0) if rng.random() < .5 else (np.nan, np.nan, np.nan)
            it = int(rng.integers(1, 12))
            a = (it, float(rng.choice([0.0, 0.3])), float(rng.choice([-3e38, 0.01])))
            got = ctx.icp_cloud_to_cloud_ex(ref, model, *a, L.FL_ICP_PARITY)
            exp = O.icp(ref, model, *a, accum64=False, use_kdtree=True)
            ok = got["iters"] == exp["iters"] and got["n_corr_last"] == exp["n_corr_last"] and \
                np.array_equal(bits(got["R"]), bits(exp["R"])) and np.array_equal(bits(got["T"]), bits(exp["T"]))
            stats["icp"] += 1
        elif kind == 1:    # front-end on random size images
            w, h = int(rng.integers(16, 400)), int(rng.integers(16, 300))
            bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            if rng.random() < .5: bgr = (bgr // 32 * 32).astype(np.uint8)
            dep = (600 + rng.integers(0, int(rng.integers(2, 400)), (h, w))).astype(np.uint16)
            dep[rng.random((h, w)) < 0.02] = 0
            ok = np.array_equal(ctx.quantized_orientations(bgr, 10.0), O.quantized_orientations(bgr, 10.0)) and \
                np.array_equal(ctx.quantized_normals(dep), O.quantized_normals(dep)) and \
                (w < 2 or h < 2 or np.array_equal(ctx.pyrdown_bgr(bgr), O.pyrdown_bgr(bgr)))
            dw, dh = int(rng.integers(8, 400)), int(rng.integers(8, 300))
            ok = ok and np.array_equal(ctx.resize_linear(bgr, dw, dh), O.resize_linear_u8(bgr, dw, dh)) and \
                np.array_equal(ctx.resize_linear(dep, dw, dh), O.resize_linear_u16(dep, dw, dh))
            stats["frontend"] += 1
        elif kind == 2:    # whole Recognition (lazy fine levels, poisoned outside the marked tiles) on random scenes
            lv = int(rng.integers(2, 4))
            T = [[5, 8], [5, 8, 4]][lv - 2]
            sc = synth.recognition_scene(lambda b, d, l: O.quantize_pyramid(b, d, l), levels=lv, seed=seed, n_views=3, n_random=10)
            det = api.Detector(ctx, 2, T); det.add_class(sc["bank"]); det.finalize(640, 480, max_batch=2)
            it = int(rng.integers(1, 12))
            thr = float(rng.choice([60.0, 70.0, 80.0]))
            ax, sh = int(rng.integers(0, 2)), int(rng.integers(-300, 300))
            fb = [sc["bgr"], np.roll(sc["bgr"], sh, axis=ax)]
            fd = [sc["depth"], np.roll(sc["depth"], sh, axis=ax)]
            # the caller's intrinsics need not be the camera's: the organised search projects with whatever it is given
            K = tuple(float(v) for v in (np.array(sc["K"]) * rng.uniform(0.97, 1.03, 4))) if rng.random() < 0.5 else sc["K"]
            rs = det.recognize_batch(fb, fd, K, thr, it, 0.3, 0.01)
            ok = True
            for b_, d_, r in zip(fb, fd, rs):
                e = O.recognition(b_, d_, K, T, sc["bank"], thr, it, 0.3, 0.01)
                ok = ok and r["found"] == e["found"] and r["n_matches"] == e["n_matches"] and \
                    (not e["found"] or np.array_equal(bits(r["pose"]), bits(e["pose"])))
            det.close()
            stats["recognition"] += 1
        elif kind == 5:    # detection(): random crop rectangles, poses, intrinsics and iteration counts (organised search, 1024-thread kernel)
            R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)), tz=float(rng.uniform(560, 800)),
                                     yaw=float(rng.uniform(-0.5, 0.5)), tilt=float(rng.uniform(0.2, 0.5)), roll=float(rng.uniform(-0.2, 0.2)))
            scene, _, ms = synth.render(640, 480, R, t, seed=seed, noise=True, background=bool(rng.integers(0, 2)))
            dR = synth.rot_z(rng.uniform(-.06, .06)) @ synth.rot_x(rng.uniform(-.05, .05)) @ synth.rot_y(rng.uniform(-.05, .05))
            tm = t + rng.uniform(-25, 25, 3) * np.array([1, 1, 0.3])
            model, _, mm = synth.render(640, 480, dR @ R, tm, seed=seed + 7, noise=False, background=False)
            ys, xs = np.nonzero(ms); ym, xm = np.nonzero(mm)
            cw = int(min(max(xs.max() - xs.min(), xm.max() - xm.min()) + rng.integers(0, 12), 600))
            ch = int(min(max(ys.max() - ys.min(), ym.max() - ym.min()) + rng.integers(0, 12), 440))
            rr = (int(np.clip(xs.min() - rng.integers(0, 6), 0, 640 - cw)), int(np.clip(ys.min() - rng.integers(0, 6), 0, 480 - ch)), cw, ch)
            rm = (int(np.clip(xm.min() - rng.integers(0, 6), 0, 640 - cw)), int(np.clip(ym.min() - rng.integers(0, 6), 0, 480 - ch)), cw, ch)
            K = tuple(float(v) for v in (np.array([608.0, 608.0, 320.0, 240.0]) * rng.uniform(0.95, 1.05, 4)))
            a = (model, scene, K, rm, rr, int(rng.integers(1, 14)), float(rng.choice([0.0, 0.3])), float(rng.choice([-3e38, 0.01])),
                 (dR @ R).astype(np.float32), tm.astype(np.float32))
            g = ctx.detection(*a, L.FL_ICP_PARITY)
            e = O.detection(*a)
            ok = g["n_points"] == e["n_points"] and g["icp"]["iters"] == e["icp"]["iters"] and \
                np.array_equal(bits(g["R_final"]), bits(e["R_final"])) and np.array_equal(bits(g["T_final"]), bits(e["T_final"]))
            stats["detection"] += 1
        elif kind == 4:    # template extraction (addTemplate) on a random view, with / without mask, 1-3 levels
            R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)), tz=float(rng.uniform(560, 760)),
                                     yaw=float(rng.uniform(-0.5, 0.5)), tilt=float(rng.uniform(0.2, 0.5)), roll=float(rng.uniform(-0.2, 0.2)))
            depth, bgr, mask = synth.render(640, 480, R, t, seed=seed, noise=bool(rng.integers(0, 2)), background=True)
            mk = (mask * 255).astype(np.uint8) if rng.random() < 0.7 else None
            lv = int(rng.integers(1, 4))
            e = O.add_template(bgr, depth, mk, lv)
            g = ctx.extract_template_pyramid(bgr, depth, mk, lv)
            ok = (e is None) == (g is None)
            if ok and e is not None:
                ok = tuple(g[1]) == tuple(e[2]) and all(
                    np.array_equal(g[0][k]["features"], np.stack([e[1][k]["x"], e[1][k]["y"], e[1][k]["label"]], 1)) and
                    g[0][k]["width"] == int(e[0][k]["width"]) and g[0][k]["offset_y"] == int(e[0][k]["offset_y"]) for k in range(lv * 2))
            stats["extract"] += 1
        else:              # LINEMOD on random quantized pyramids / banks
            levels = int(rng.integers(1, 4))
            T = [[8], [4, 8], [4, 8, 4]][levels - 1]
            w0, h0 = [(160, 128), (320, 160), (320, 256)][levels - 1]
            dens = float(rng.uniform(0.02, 0.08))
            qs = [synth.random_quantized(rng, w0 >> l, h0 >> l, dens) for l in range(levels) for _ in range(2)]
            bank = synth.make_bank("o", int(rng.integers(1, 40)), levels, 2, w0, h0, seed=seed, qs=qs, planted_frac=0.3, bbox=64)
            det = api.Detector(ctx, 2, T); det.add_class(bank); det.finalize(w0, h0)
            thr = float(rng.uniform(50, 90))
            got, ng = det.match_quantized(qs, thr)
            exp, ne = O.match_quantized(qs, w0, h0, T, [bank], thr)
            ok = ng == ne and all(np.array_equal(got[k], exp[k]) for k in ("x", "y", "class_idx", "template_id")) and \
                np.array_equal(got["similarity"].view(np.uint32), exp["similarity"].view(np.uint32))
            det.close()
            stats["linemod"] += 1
    except Exception as ex:      # noqa: BLE001
        ok = False
        print("seed", seed, "kind", kind, "raised", repr(ex)[:300])
    if not ok:
        stats["fail"] += 1
        print("MISMATCH seed", seed, "kind", kind)
print("fuzz done:", stats, "in %.0f s" % (time.time() - t0))
sys.exit(1 if stats["fail"] else 0)
