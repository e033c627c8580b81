"""GPU parity: back-projection, ICP, detection() and Recognition() vs the oracle.

FL_ICP_PARITY accumulates the reference's float32 sums as sequential chains, so every number is
expected to be bit-identical to the oracle's float32 mode (tolerance 0 is asserted where the
whole chain is deterministic; the pose bar from BASELINE.json's north_star is 1e-4).
FL_ICP_FAST is compared with the oracle's fp64-accumulation yardstick.
"""
import numpy as np
import pytest

from fealess_amd import api, synth
from fealess_amd import _lib as L

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4          # north_star: "within 1e-4 on ICP's final 4x4 pose"


def _clouds(seed, n=6000, noise=0.3):
    """A paired cloud: reference = points on the synthetic object, model = rigidly perturbed copy."""
    rng = np.random.default_rng(seed)
    R, t = synth.object_pose(tz=650.0)
    depth, _, mask = synth.render(640, 480, R, t, seed=seed, noise=True, background=False)
    ys, xs = np.nonzero(mask)
    sel = rng.choice(len(ys), size=min(n, len(ys)), replace=False)
    sel.sort()
    z = depth[ys[sel], xs[sel]].astype(np.float32)
    ref = np.stack([(xs[sel] - 320.0) / 608.0 * z, (ys[sel] - 240.0) / 608.0 * z, z], 1).astype(np.float32)
    dR = synth.rot_z(0.02) @ synth.rot_x(-0.015) @ synth.rot_y(0.01)
    c = ref.mean(0)
    model = ((ref - c) @ dR.T + c + np.array([1.5, -2.0, 1.0])).astype(np.float32)
    model += rng.normal(0, noise, model.shape).astype(np.float32)
    return ref, model


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_wave_reductions_of_the_search_step_agree(ctx):
    """The six whole-wave maxima a search step takes (union rectangle, tallest and widest lane window) are folded through
    gfx950's lane-swap instructions (wave_max6, fl_icp.hip); the plain DPP reduction (wave_max_multi) and numpy must agree
    on every value, including ties, extremes and values confined to one lane."""
    import ctypes as C
    lib = L.load()
    lib.fl_dev_wave_max6.restype = C.c_int
    lib.fl_dev_wave_max6.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(7)
    cases = [rng.integers(-2 ** 31, 2 ** 31 - 1, size=(6, 64), dtype=np.int64).astype(np.int32) for _ in range(20)]
    cases += [rng.integers(-40, 700, size=(6, 64)).astype(np.int32) for _ in range(20)]
    for lane in (0, 15, 16, 31, 32, 47, 48, 63):             # the maximum of value k sits in one lane only
        a = np.full((6, 64), -0x3fffffff, np.int32)
        a[:, lane] = np.arange(6) * 100 + lane
        cases.append(a)
    a = np.arange(6 * 64, dtype=np.int32).reshape(6, 64) * np.array([1, -1, 3, -3, 7, -7], np.int32)[:, None]
    cases.append(a)
    for a in cases:
        a = np.ascontiguousarray(a, np.int32)
        out = np.zeros(12, np.int32)
        rc = lib.fl_dev_wave_max6(a.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0
        want = a.max(axis=1)
        assert np.array_equal(out[:6], want), (out[:6], want)
        assert np.array_equal(out[6:], want), (out[6:], want)


@pytest.mark.parametrize("w,h,K", [(640, 480, (608.0, 608.0, 320.0, 240.0)), (1280, 720, (915.3, 917.1, 641.2, 358.7)),
                                   (33, 17, (50.0, 60.0, 16.0, 8.0))])
def test_depth_to_3d_bit_exact(ctx, oracle, w, h, K):
    depth = np.random.default_rng(w).integers(0, 3000, (h, w)).astype(np.uint16)
    depth[0, :5] = 0
    got = ctx.depth_to_3d(depth, *K)
    exp = oracle.depth_to_3d(depth, *K)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.array_equal(_bits(np.nan_to_num(got)), _bits(np.nan_to_num(exp)))


@pytest.fixture(params=["1024", "256", "256x5"])
def width(request, ctx):
    """Every build of the ICP kernels: the library picks the 1024-thread workgroup for batches of up to two jobs per CU (with the
    search running ahead of the dist_mean chain) and a 256-thread one beyond, compiled for 4 or for 5 workgroups per CU (whichever
    finishes the batch sooner); the options icp_wide and icp_occ (fl_context_set_option) force any of them for any batch."""
    ctx.set_option("icp_wide", 1 if request.param == "1024" else 0)
    ctx.set_option("icp_occ", 5 if request.param == "256x5" else 0)
    yield request.param
    ctx.set_option("icp_wide", -1)
    ctx.set_option("icp_occ", 0)


@pytest.mark.parametrize("seed,n,it", [(1, 6000, 20), (2, 1500, 10), (3, 12000, 6)])
def test_icp_parity_mode_matches_oracle32(ctx, oracle, seed, n, it, width):
    ref, model = _clouds(seed, n)
    got = ctx.icp_cloud_to_cloud_ex(ref, model, it, 0.0, -3.0e38, L.FL_ICP_PARITY)
    exp = oracle.icp(ref, model, it, 0.0, -3.0e38, accum64=False, use_kdtree=True)
    assert got["iters"] == exp["iters"] == it
    assert got["n_corr_last"] == exp["n_corr_last"]
    assert np.abs(got["R"] - exp["R"]).max() <= POSE_TOL
    assert np.abs(got["T"] - exp["T"]).max() <= POSE_TOL * max(1.0, np.abs(exp["T"]).max())
    # the chains are deterministic: in practice every bit agrees
    assert np.array_equal(_bits(got["R"]), _bits(exp["R"])) and np.array_equal(_bits(got["T"]), _bits(exp["T"]))
    assert _bits(got["dist_mean"]) == _bits(exp["dist_mean"])


def test_icp_default_thresholds_early_exit(ctx, oracle, width):
    ref, model = _clouds(4, 5000)
    got = ctx.icp_cloud_to_cloud_ex(ref, model, 10, 0.5, 0.01, L.FL_ICP_PARITY)
    exp = oracle.icp(ref, model, 10, 0.5, 0.01)
    assert got["iters"] == exp["iters"]
    assert np.array_equal(_bits(got["R"]), _bits(exp["R"])) and np.array_equal(_bits(got["T"]), _bits(exp["T"]))
    assert _bits(got["px_ratio"]) == _bits(exp["px_ratio"])


def _pose_dist(a, b, scale):
    """(max |dR|, max |dT| / scale): the north_star's "within 1e-4 on ICP's final 4x4 pose" read as absolute on the rotation
    entries and relative on the translation -- relative to `scale`, the size of the coordinates the pose acts on (mm)."""
    dr = float(np.abs(np.asarray(a["R"], np.float64) - np.asarray(b["R"], np.float64)).max())
    dt = float(np.abs(np.asarray(a["T"], np.float64) - np.asarray(b["T"], np.float64)).max() / scale)
    return dr, dt


def test_icp_fast_mode_close_to_fp64_yardstick(ctx, oracle):
    """FL_ICP_FAST against the two things it can be compared with: the oracle's fp64-accumulation yardstick (what the sums
    would be without float32 summation noise) and the oracle's float32 mode = the reference's own arithmetic, the bar the
    north_star's 1e-4 is stated against (ICP.cpp:8-25,731-735: sequential float32 sums).  |f32 - f64| is the reference's
    own summation noise floor (measured on the box: R 1.2e-5, T 1.6e-3 mm); FAST cannot be closer to f32 than that floor
    (it sits 5e-7 / 7e-5 mm from the exact sums), and must be within 1e-4 of both, T relative to the clouds' size."""
    worst = [0.0] * 6
    for seed, n in ((5, 6000), (6, 9000), (7, 3000)):
        ref, model = _clouds(seed, n)
        got = ctx.icp_cloud_to_cloud_ex(ref, model, 20, 0.0, -3.0e38, L.FL_ICP_FAST)
        e64 = oracle.icp(ref, model, 20, 0.0, -3.0e38, accum64=True)
        e32 = oracle.icp(ref, model, 20, 0.0, -3.0e38, accum64=False)
        assert got["iters"] == e64["iters"] == e32["iters"]
        scale = float(np.abs(ref).max())                        # ~ 700 mm: the clouds' coordinates
        d = _pose_dist(got, e64, scale) + _pose_dist(got, e32, scale) + _pose_dist(e32, e64, scale)
        worst = [max(a, b) for a, b in zip(worst, d)]
        assert np.abs(got["T"] - e64["T"]).max() <= 1e-3          # mm, against the exact sums
    print("FL_ICP_FAST on clouds: |FAST-f64| R %.3g T(rel) %.3g; |FAST-f32| R %.3g T(rel) %.3g; |f32-f64| R %.3g T(rel) %.3g" % tuple(worst))
    assert worst[0] <= POSE_TOL and worst[1] <= POSE_TOL          # vs the fp64 yardstick
    assert worst[2] <= POSE_TOL and worst[3] <= POSE_TOL          # vs the reference's float32 arithmetic (north_star's bar)


def test_icp_fast_mode_recognition_vs_the_f32_oracle(ctx, oracle):
    """The same question for the whole Recognition() (crop back-projection, pre-alignment, ICP, pose composition): the
    final 4x4 of FL_ICP_FAST against the float32 oracle's (what the reference computes) and against the fp64 yardstick."""
    worst = [0.0] * 6
    for seed in (3, 4, 5):
        sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=seed, n_views=3)
        det = api.Detector(ctx, 2, [5, 8])
        det.add_class(sc["bank"])
        det.finalize(640, 480, max_batch=1)
        got = det.recognize_batch([sc["bgr"]], [sc["depth"]], sc["K"], 75.0, 20, 0.0, -3.0e38, mode=L.FL_ICP_FAST)[0]
        det.close()
        e32 = oracle.recognition(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], 75.0, 20, 0.0, -3.0e38, accum64=False)
        e64 = oracle.recognition(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], 75.0, 20, 0.0, -3.0e38, accum64=True)
        assert got["found"] == e32["found"] == e64["found"] == 1
        assert got["best"]["template_id"] == e32["best"]["template_id"]

        def pose(r):
            return dict(R=r["pose"][:3, :3], T=r["pose"][:3, 3])
        scale = float(np.abs(e32["pose"][:3, 3]).max())           # ~ 650 mm: the object's distance
        d = _pose_dist(pose(got), pose(e64), scale) + _pose_dist(pose(got), pose(e32), scale) + _pose_dist(pose(e32), pose(e64), scale)
        worst = [max(a, b) for a, b in zip(worst, d)]
    print("FL_ICP_FAST Recognition: |FAST-f64| R %.3g T(rel) %.3g; |FAST-f32| R %.3g T(rel) %.3g; |f32-f64| R %.3g T(rel) %.3g" % tuple(worst))
    assert worst[0] <= POSE_TOL and worst[1] <= POSE_TOL
    # On these 15 k-point clouds the reference's own float32 summation noise |f32 - f64| is 1.1e-4 (R) / 1.1e-4 (T relative),
    # i.e. already past the north_star's 1e-4: FAST (5e-6 / 1e-6 from the exact sums) can be no closer to the float32 result
    # than that floor.  The honest bar: FAST is within 1e-4 of the exact sums, and no farther from the reference's float32
    # result than the reference is from the exact sums (+ FAST's own distance to them); FL_ICP_PARITY is the mode that
    # reproduces the float32 result bit for bit.
    assert worst[2] <= worst[4] + worst[0] + 1e-7 and worst[3] <= worst[5] + worst[1] + 1e-7
    assert worst[2] <= 2.5 * POSE_TOL and worst[3] <= 2.5 * POSE_TOL


def test_icp_edge_cases(ctx, oracle, width):
    ref, model = _clouds(6, 200)
    r = ctx.icp_cloud_to_cloud_ex(ref[:2], model[:2], 5)
    assert r["dist_mean"] == -1.0 and not r["R"].any() and r["iters"] == 0      # < 3 points (ICP.cpp:633-638)
    # invalid (z > 900) and NaN points inside the clouds
    ref2, model2 = ref.copy(), model.copy()
    model2[5] = (1.0, 2.0, 950.0)
    ref2[9] = (3.0, 4.0, 1000.0)
    model2[11] = (np.nan, np.nan, np.nan)
    got = ctx.icp_cloud_to_cloud_ex(ref2, model2, 8, 0.0, -3.0e38)
    exp = oracle.icp(ref2, model2, 8, 0.0, -3.0e38)
    assert got["iters"] == exp["iters"] and got["n_corr_last"] == exp["n_corr_last"]
    assert np.array_equal(_bits(got["R"]), _bits(exp["R"])) and np.array_equal(_bits(got["T"]), _bits(exp["T"]))
    # far-apart clouds: no correspondence survives after iteration 1 -> iter jumps to the limit
    far = model + np.float32(500.0)
    far[:, 2] = model[:, 2]
    got = ctx.icp_cloud_to_cloud_ex(ref, far, 6, 0.0, -3.0e38)
    exp = oracle.icp(ref, far, 6, 0.0, -3.0e38)
    assert got["iters"] == exp["iters"] and got["n_corr_last"] == exp["n_corr_last"]
    assert np.abs(got["R"] - exp["R"]).max() <= POSE_TOL


def test_icp_unequal_sizes(ctx, oracle):
    ref, model = _clouds(7, 3000)
    got = ctx.icp_cloud_to_cloud_ex(ref, model[:2500], 6, 0.0, -3.0e38)
    exp = oracle.icp(ref, model[:2500], 6, 0.0, -3.0e38)
    assert got["iters"] == exp["iters"]
    assert np.array_equal(_bits(got["R"]), _bits(exp["R"])) and np.array_equal(_bits(got["T"]), _bits(exp["T"]))


def test_detection_matches_oracle(ctx, oracle, width):
    R, t = synth.object_pose(tx=10, ty=-5, tz=660)
    scene, _, _ = synth.render(640, 480, R, t, seed=3)
    R2 = synth.rot_z(0.03) @ R
    model, _, _ = synth.render(640, 480, R2, t + np.array([25.0, 15.0, 6.0]), seed=4, noise=False, background=False)
    rect_model = (230, 150, 180, 150)
    rect_ref = (215, 140, 180, 150)
    K = (608.0, 608.0, 320.0, 240.0)
    rm = np.eye(3, dtype=np.float32)
    tm = np.array([1.0, 2.0, 3.0], np.float32)
    for mode, acc in ((L.FL_ICP_PARITY, False), (L.FL_ICP_FAST, True)):
        got = ctx.detection(model, scene, K, rect_model, rect_ref, 20, 0.0, -3.0e38, rm, tm, mode)
        exp = oracle.detection(model, scene, K, rect_model, rect_ref, 20, 0.0, -3.0e38, rm, tm, accum64=acc)
        assert got["n_points"] == exp["n_points"] > 1000
        assert got["icp"]["iters"] == exp["icp"]["iters"]
        assert np.abs(got["R_final"] - exp["R_final"]).max() <= POSE_TOL
        if acc:
            assert np.abs(got["T_final"] - exp["T_final"]).max() <= 1e-3
        else:
            assert np.array_equal(_bits(got["T_final"]), _bits(exp["T_final"]))
            assert np.array_equal(_bits(got["R_final"]), _bits(exp["R_final"]))
    with pytest.raises(api.FealessError) as e:      # Q10: rect outside the image
        ctx.detection(model, scene, K, (600, 400, 180, 150), (600, 400, 180, 150), 5, 0.0, 0.0, rm, tm)
    assert e.value.code == -3


@pytest.mark.parametrize("seed", [3, 8])
def test_recognition_matches_oracle(ctx, oracle, seed, width):
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=seed, n_views=5,
                                 n_random=30)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=3)
    frames_b = [sc["bgr"], sc["bgr"][:, ::-1].copy(), sc["bgr"]]
    frames_d = [sc["depth"], sc["depth"][:, ::-1].copy(), sc["depth"]]
    for params in ((75.0, 10, 0.5, 0.01), (75.0, 20, 0.0, -3.0e38)):
        got = det.recognize_batch(frames_b, frames_d, sc["K"], *params)
        for i in range(3):
            exp = oracle.recognition(frames_b[i], frames_d[i], sc["K"], [5, 8], sc["bank"], *params)
            g = got[i]
            assert g["status"] == 0 and g["found"] == exp["found"], i
            assert g["n_matches"] == exp["n_matches"]
            if not exp["found"]:
                continue
            for k in ("x", "y", "template_id"):
                assert g["best"][k] == exp["best"][k]
            assert g["best"]["similarity"] == exp["best"]["similarity"]
            assert g["det"]["n_points"] == exp["det"]["n_points"]
            assert g["det"]["icp"]["iters"] == exp["det"]["icp"]["iters"]
            assert np.abs(g["pose"][:3, :3] - exp["pose"][:3, :3]).max() <= POSE_TOL
            assert np.abs(g["pose"][:3, 3] - exp["pose"][:3, 3]).max() <= POSE_TOL * 700
            assert np.array_equal(_bits(g["pose"]), _bits(exp["pose"]))
        assert got[0]["found"] == 1
    det.close()


@pytest.mark.parametrize("scale", [1e-3, 37.0])
def test_icp_is_scale_free(ctx, oracle, scale):
    """Clouds in metres (or any other unit): the search grid adapts its cell size, results stay bit-exact."""
    ref, model = _clouds(8, 4000)
    ref, model = (ref * np.float32(scale)).astype(np.float32), (model * np.float32(scale)).astype(np.float32)
    if scale > 1:                        # keep z under the reference's validity bound (z <= 900, common.cpp:261-266)
        ref[:, 2] -= np.float32(600.0 * scale - 100.0)
        model[:, 2] -= np.float32(600.0 * scale - 100.0)
    got = ctx.icp_cloud_to_cloud_ex(ref, model, 8, 0.0, -3.0e38)
    exp = oracle.icp(ref, model, 8, 0.0, -3.0e38)
    assert got["iters"] == exp["iters"] == 8 and got["n_corr_last"] == exp["n_corr_last"]
    assert np.array_equal(_bits(got["R"]), _bits(exp["R"])) and np.array_equal(_bits(got["T"]), _bits(exp["T"]))


def test_icp_degenerate_clouds(ctx, oracle):
    """Collinear and coincident reference points (zero-area bounding box) do not break the grid."""
    rng = np.random.default_rng(5)
    t = np.sort(rng.uniform(0, 100, 500)).astype(np.float32)
    ref = np.stack([t, np.full_like(t, 3.0), np.full_like(t, 400.0)], 1)
    model = (ref + np.float32([0.3, 0.0, 0.2])).astype(np.float32)
    for r, m in ((ref, model), (np.repeat(ref[:1], 50, 0), np.repeat(model[:1], 50, 0))):
        got = ctx.icp_cloud_to_cloud_ex(r, m, 4, 0.0, -3.0e38)
        exp = oracle.icp(r, m, 4, 0.0, -3.0e38)
        assert got["iters"] == exp["iters"] and got["n_corr_last"] == exp["n_corr_last"]
        assert np.array_equal(np.isnan(got["R"]), np.isnan(exp["R"]))
        assert np.abs(np.nan_to_num(got["R"]) - np.nan_to_num(exp["R"])).max() <= POSE_TOL


def test_multi_hypothesis_topk_and_nms(ctx, oracle):
    """the first k matches refined in one launch (k ICP workgroups) + nonMaximumSuppression
    (ICP/NMS.cpp:6-40): every hypothesis equals the oracle's refinement of the same match, bit for bit."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=21, n_views=6)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480)
    k = 6
    got = det.recognize_topk(sc["bgr"], sc["depth"], sc["K"], k, 60.0, 8, 0.3, 0.01)
    exp, win_exp = oracle.recognition_topk(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], k, 60.0, 8, 0.3, 0.01, nms_dist=20.0)
    assert len(got) == len(exp) >= 2
    for g, e in zip(got, exp):
        assert g["found"] == e["found"] and g["best"]["template_id"] == e["best"]["template_id"]
        assert (g["best"]["x"], g["best"]["y"]) == (e["best"]["x"], e["best"]["y"])
        if e["found"]:
            assert np.array_equal(_bits(g["pose"]), _bits(e["pose"])) and g["det"]["n_points"] == e["det"]["n_points"]
            assert _bits(g["det"]["icp"]["dist_mean"]) == _bits(e["det"]["icp"]["dist_mean"])
    # hypothesis 0 is what Recognition() returns
    r0 = det.recognize_batch([sc["bgr"]], [sc["depth"]], sc["K"], 60.0, 8, 0.3, 0.01)[0]
    assert np.array_equal(_bits(r0["pose"]), _bits(got[0]["pose"]))
    assert det.nms(len(got), 20.0) == win_exp and len(win_exp) >= 1
    assert det.nms(len(got), 0.0) == list(range(len(got)))          # nothing is closer than 0: every hypothesis survives
    assert det.nms(len(got), 1e9) in ([0], [win_exp[0]]) and len(det.nms(len(got), 1e9)) == 1
    det.close()


# ---- FL_ICP_POINT_TO_PLANE (no reference counterpart, so no oracle) ----------------
# Yardsticks: tests/p2plane_model.py (an independent numpy/scipy statement of the same algorithm) and the
# ground-truth pose of the synthetic scene.
def _angle_deg(Ra, Rb):
    """Rotation angle of Ra Rb^T from its skew part and its trace (atan2): accurate for small angles, where the
    arccos of a trace made of float32-rounded entries resolves nothing below 0.03 degrees."""
    M = np.asarray(Ra, np.float64) @ np.asarray(Rb, np.float64).T
    s = 0.5 * np.linalg.norm([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    return float(np.degrees(np.arctan2(s, (np.trace(M) - 1) / 2)))


def _bbox(m):
    ys, xs = np.nonzero(m)
    return int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1


def _plane_case(seed, da, dt):
    Rs, ts = synth.object_pose(tx=12, ty=-8, tz=655)
    scene, _, ms = synth.render(640, 480, Rs, ts, seed=10 + seed)
    Rm = synth.rot_z(da[0]) @ synth.rot_x(da[1]) @ synth.rot_y(da[2]) @ Rs
    tm = ts + np.array(dt, float) + np.array([20.0, -12.0, 0.0])
    model, _, mm = synth.render(640, 480, Rm, tm, seed=20 + seed, noise=False, background=False)
    bs, bm = _bbox(ms), _bbox(mm)
    cw = max(bs[2] - bs[0], bm[2] - bm[0]) + 8
    ch = max(bs[3] - bs[1], bm[3] - bm[1]) + 8
    return dict(scene=scene, model=model, Rs=Rs, ts=ts, Rm=Rm.astype(np.float32), tm=tm.astype(np.float32),
                rect_ref=(bs[0] - 4, bs[1] - 4, cw, ch), rect_model=(bm[0] - 4, bm[1] - 4, cw, ch))


PLANE_CASES = [((0.05, 0.03, -0.04), (6, -4, 5)), ((0.03, -0.05, 0.02), (-5, 3, -6)), ((-0.06, 0.02, 0.05), (3, 6, 4))]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_point_to_plane_detection_vs_model_and_ground_truth(ctx, seed):
    import p2plane_model as P
    c = _plane_case(seed, *PLANE_CASES[seed])
    K = (608.0, 608.0, 320.0, 240.0)
    args = (c["model"], c["scene"], K, c["rect_model"], c["rect_ref"], 20, 0.0, -3.0e38, c["Rm"], c["tm"])
    got = ctx.detection(*args, L.FL_ICP_POINT_TO_PLANE)
    exp = P.detection_point_to_plane(*args)
    assert got["n_points"] == exp["n_points"] > 5000
    assert got["icp"]["iters"] == exp["icp"]["iters"] == 20
    # the kernel against the independent model: same pairs, same sums up to fp rounding
    assert abs(got["icp"]["n_corr_last"] - exp["icp"]["n_corr_last"]) <= 5
    assert _angle_deg(got["R_final"], exp["R_final"]) <= 0.02
    assert np.abs(got["T_final"] - exp["T_final"]).max() <= 0.05
    assert abs(got["icp"]["dist_mean"] - exp["icp"]["dist_mean"]) <= 1e-2
    # both against the truth, and against the reference's point-to-point ICP on the same input
    p2p = ctx.detection(*args, L.FL_ICP_PARITY)
    e_plane, e_p2p = _angle_deg(got["R_final"], c["Rs"]), _angle_deg(p2p["R_final"], c["Rs"])
    assert e_plane <= 1.0 and e_plane <= 0.5 * e_p2p, (e_plane, e_p2p)
    assert np.linalg.norm(got["T_final"] - c["ts"]) <= 0.5
    assert np.linalg.norm(got["T_final"] - c["ts"]) <= np.linalg.norm(p2p["T_final"] - c["ts"])


def test_point_to_plane_on_clouds_and_argument_checks(ctx):
    import p2plane_model as P
    c = _plane_case(1, *PLANE_CASES[1])
    K = (608.0, 608.0, 320.0, 240.0)
    ref, mod, sx, sy = P.crop_pairs(c["model"], c["scene"], K, c["rect_model"], c["rect_ref"])
    nrm = P.scene_normals(c["scene"], K, sx, sy)
    assert (np.abs(np.linalg.norm(nrm, axis=1) - 1) < 1e-5).mean() > 0.8       # most points have a normal
    mod = (mod + (ref.mean(0) - mod.mean(0))).astype(np.float32)
    got = ctx.icp_point_to_plane(ref, nrm, mod, 15, 0.0, -3.0e38)
    exp = P.icp_point_to_plane(ref, nrm, mod, 15, 0.0, -3.0e38)
    assert got["iters"] == exp["iters"] == 15
    assert _angle_deg(got["R"], exp["R"]) <= 0.02 and np.abs(got["T"] - exp["T"]).max() <= 0.2
    # default thresholds stop early, exactly like icpCloudToCloud_Ex's loop control
    got = ctx.icp_point_to_plane(ref, nrm, mod, 50, 0.5, 0.01)
    exp = P.icp_point_to_plane(ref, nrm, mod, 50, 0.5, 0.01)
    assert got["iters"] == exp["iters"] < 50
    # all-zero normals constrain nothing: every iteration is skipped (counted), the pose stays the identity
    r = ctx.icp_point_to_plane(ref, np.zeros_like(nrm), mod, 4, 0.0, -3.0e38)
    assert r["iters"] == 4 and np.array_equal(r["R"], np.eye(3, dtype=np.float32)) and not r["T"].any()
    # fl_icp has no normals to offer
    with pytest.raises(api.FealessError) as e:
        ctx.icp_cloud_to_cloud_ex(ref, mod, 4, 0.0, 0.0, L.FL_ICP_POINT_TO_PLANE)
    assert e.value.code == L.FL_ERR_INVALID
    with pytest.raises(api.FealessError):
        ctx.icp_cloud_to_cloud_ex(ref, mod, 4, 0.0, 0.0, 7)
    r = ctx.icp_point_to_plane(ref[:2], nrm[:2], mod[:2], 4)
    assert r["dist_mean"] == -1.0 and r["iters"] == 0


def test_point_to_plane_recognition_improves_the_pose(ctx, oracle):
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=3, n_views=5, n_random=10)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=2)
    frames_b, frames_d = [sc["bgr"], sc["bgr"]], [sc["depth"], sc["depth"]]
    p2p = det.recognize_batch(frames_b, frames_d, sc["K"], 75.0, 20, 0.0, -3.0e38, L.FL_ICP_PARITY)
    pl = det.recognize_batch(frames_b, frames_d, sc["K"], 75.0, 20, 0.0, -3.0e38, L.FL_ICP_POINT_TO_PLANE)
    for a, b in zip(p2p, pl):
        assert a["found"] == b["found"] == 1 and a["best"] == b["best"]        # the LINEMOD half is untouched
        ea, eb = _angle_deg(a["pose"][:3, :3], sc["R_true"]), _angle_deg(b["pose"][:3, :3], sc["R_true"])
        assert eb <= max(1.0, 0.75 * ea), (ea, eb)
        assert np.linalg.norm(b["pose"][:3, 3] - sc["t_true"]) <= max(1.0, np.linalg.norm(a["pose"][:3, 3] - sc["t_true"]))
    assert np.array_equal(pl[0]["pose"], pl[1]["pose"])
    det.close()


# ---- lazy fine levels: fl_recognize_* quantise / spread the finer levels only in the tiles the candidates touch ----
@pytest.mark.parametrize("levels,T", [(2, [5, 8]), (3, [5, 8, 4])])
def test_lazy_fine_levels_equal_eager_and_oracle(ctx, oracle, levels, T):
    """Same results with the finer levels computed lazily (default) and eagerly (option eager_frontend = 1), for frames
    whose object sits mid-image, is pushed against each border (patches that leave their linear memory mark the whole
    frame) or is absent; conftest.py poisons everything outside the marked tiles."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=levels, seed=11, n_views=5, n_random=25)
    frames_b, frames_d = [sc["bgr"]], [sc["depth"]]
    for sh in (-200, 230):
        frames_b.append(np.roll(sc["bgr"], sh, axis=1)); frames_d.append(np.roll(sc["depth"], sh, axis=1))
    for sh in (-150, 170):
        frames_b.append(np.roll(sc["bgr"], sh, axis=0)); frames_d.append(np.roll(sc["depth"], sh, axis=0))
    frames_b.append(np.full_like(sc["bgr"], 90)); frames_d.append(np.full_like(sc["depth"], 1200))
    results = {}
    for mode in ("lazy", "eager"):
        ctx.set_option("eager_frontend", 1 if mode == "eager" else 0)      # sampled by fl_detector_finalize
        det = api.Detector(ctx, 2, T)
        det.add_class(sc["bank"])
        det.finalize(640, 480, max_batch=len(frames_b))
        ctx.set_option("eager_frontend", 0)
        results[mode] = det.recognize_batch(frames_b, frames_d, sc["K"], 70.0, 10, 0.5, 0.01)
        t = det.stage_times()
        assert (t["lazy_frontend_ms"] > 0) == (mode == "lazy")
        if mode == "lazy":
            with pytest.raises(api.FealessError):            # no whole quantised pyramid after a lazy batch
                det.last_quantized()
        det.close()
    n_found = 0
    for i, (a, b) in enumerate(zip(results["lazy"], results["eager"])):
        exp = oracle.recognition(frames_b[i], frames_d[i], sc["K"], T, sc["bank"], 70.0, 10, 0.5, 0.01)
        for g in (a, b):
            assert g["status"] == 0 and g["found"] == exp["found"] and g["n_matches"] == exp["n_matches"], i
            if exp["found"]:
                assert g["best"]["x"] == exp["best"]["x"] and g["best"]["y"] == exp["best"]["y"]
                assert g["best"]["template_id"] == exp["best"]["template_id"] and g["best"]["similarity"] == exp["best"]["similarity"]
                assert np.array_equal(_bits(g["pose"]), _bits(exp["pose"]))
        n_found += exp["found"]
    assert n_found >= 2


def test_batch_topk_equals_per_frame_topk(ctx, oracle):
    """fl_recognize_batch_topk = fl_recognize_topk frame by frame (n_frames * k ICP workgroups in one launch)."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=21, n_views=6)
    frames_b = [sc["bgr"], np.roll(sc["bgr"], 40, axis=1), np.full_like(sc["bgr"], 90)]
    frames_d = [sc["depth"], np.roll(sc["depth"], 40, axis=1), np.full_like(sc["depth"], 1200)]
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=3)
    k = 4
    got = det.recognize_batch_topk(frames_b, frames_d, sc["K"], k, 60.0, 8, 0.3, 0.01)
    assert len(got) == 3 and len(got[0]) >= 2 and got[2] == []
    for f in range(3):
        one = det.recognize_topk(frames_b[f], frames_d[f], sc["K"], k, 60.0, 8, 0.3, 0.01)
        assert len(one) == len(got[f])
        for a, b in zip(one, got[f]):
            assert a["best"] == b["best"] and a["found"] == b["found"] and a["status"] == b["status"]
            assert np.array_equal(_bits(a["pose"]), _bits(b["pose"]))
            assert a["det"]["icp"]["iters"] == b["det"]["icp"]["iters"]
    det.close()


def test_device_frames_in_place_and_gathered_equal_host_frames(ctx, oracle):
    """fl_recognize_submit reads device frames at a regular pitch in place (what bench.py times), gathers device frames at
    irregular addresses into the frame workspaces, and uploads host frames on its copy stream: same results all three ways."""
    import torch
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=3, n_views=5, n_random=10)
    fb = [sc["bgr"], sc["bgr"][:, ::-1].copy(), np.roll(sc["bgr"], 60, axis=1)]
    fd = [sc["depth"], sc["depth"][:, ::-1].copy(), np.roll(sc["depth"], 60, axis=1)]
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=3)
    params = L.RecognitionParams(75.0, 10, 0.5, 0.01, L.FL_ICP_PARITY)
    host = det.recognize_batch(fb, fd, sc["K"], 75.0, 10, 0.5, 0.01)
    assert host[0]["found"] == 1
    # one device array per modality: regular pitch, read in place
    d_b = torch.from_numpy(np.stack(fb)).cuda()
    d_d = torch.from_numpy(np.stack(fd).view(np.int16)).cuda()
    torch.cuda.synchronize()
    det.recognize_submit_device([d_b.data_ptr() + i * 640 * 480 * 3 for i in range(3)],
                                [d_d.data_ptr() + i * 640 * 480 * 2 for i in range(3)], sc["K"], params)
    in_place = [api.recognition_result_to_dict(r) for r in det.recognize_collect(3)]
    # separate allocations in a different order: irregular addresses, gathered
    tb = [torch.from_numpy(fb[i]).cuda() for i in (2, 0, 1)]
    td = [torch.from_numpy(fd[i].view(np.int16)).cuda() for i in (1, 2, 0)]
    torch.cuda.synchronize()
    pb = {2: tb[0], 0: tb[1], 1: tb[2]}
    pd = {1: td[0], 2: td[1], 0: td[2]}
    det.recognize_submit_device([pb[i].data_ptr() for i in range(3)], [pd[i].data_ptr() for i in range(3)], sc["K"], params)
    gathered = [api.recognition_result_to_dict(r) for r in det.recognize_collect(3)]
    for h, a, g in zip(host, in_place, gathered):
        for o in (a, g):
            assert o["found"] == h["found"] and o["n_matches"] == h["n_matches"] and o["best"] == h["best"]
            assert np.array_equal(_bits(o["pose"]), _bits(h["pose"]))
    # back-to-back host batches alternate between the two upload buffers
    again = det.recognize_batch(fb, fd, sc["K"], 75.0, 10, 0.5, 0.01)
    third = det.recognize_batch(fb[::-1], fd[::-1], sc["K"], 75.0, 10, 0.5, 0.01)
    for h, a, t in zip(host, again, third[::-1]):
        assert np.array_equal(_bits(a["pose"]), _bits(h["pose"])) and np.array_equal(_bits(t["pose"]), _bits(h["pose"]))
    det.close()


def test_recognition_grows_the_candidate_buffers(ctx, oracle):
    """A cluttered frame (threshold low enough for thousands of coarse candidates) overflows an initial capacity of 256:
    the synchronous recognition entry points grow the buffers and run again, every frame of the batch keeps the
    oracle's result (the reference's vectors are unbounded, linemod.cpp:1490-1504)."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=5, n_views=4, n_random=20)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=2, max_candidates=256)
    thr = -100.0                               # raw threshold 0: every non-zero coarse cell of every template is a candidate
    got = det.recognize_batch([sc["bgr"], sc["bgr"]], [sc["depth"], sc["depth"]], sc["K"], thr, 6, 0.0, -3.0e38)
    exp = oracle.recognition(sc["bgr"], sc["depth"], sc["K"], [5, 8], sc["bank"], thr, 6, 0.0, -3.0e38)
    assert exp["n_matches"] > 256
    for g in got:
        assert g["status"] == 0 and g["found"] == exp["found"] == 1
        assert g["n_matches"] == exp["n_matches"]
        assert g["best"]["template_id"] == exp["best"]["template_id"] and g["best"]["similarity"] == exp["best"]["similarity"]
        assert np.array_equal(_bits(g["pose"]), _bits(exp["pose"]))
    det.close()


def test_recognition_with_2000_pyramids_matches_oracle(ctx, oracle):
    """The north_star shape's bank size: Recognition() of a few frames against a 2000-pyramid bank (rendered views + random
    pyramids) equals the oracle's, bit for bit (linemod.cpp:1458: the N-template loop is what the target is sized on)."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=21, n_views=4, n_random=1996)
    assert sc["bank"].n_pyramids == 2000
    frames_b = [sc["bgr"], np.roll(sc["bgr"], 6, axis=1), np.roll(sc["bgr"], -10, axis=1)]
    frames_d = [sc["depth"], np.roll(sc["depth"], 6, axis=1), np.roll(sc["depth"], -10, axis=1)]
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=3)
    got = det.recognize_batch(frames_b, frames_d, sc["K"], 75.0, 10, 0.5, 0.01)
    for g, b, d in zip(got, frames_b, frames_d):
        e = oracle.recognition(b, d, sc["K"], [5, 8], sc["bank"], 75.0, 10, 0.5, 0.01)
        assert g["status"] == 0 and g["found"] == e["found"] == 1 and g["n_matches"] == e["n_matches"]
        assert g["best"]["template_id"] == e["best"]["template_id"] and g["best"]["x"] == e["best"]["x"] and g["best"]["y"] == e["best"]["y"]
        assert g["best"]["similarity"] == e["best"]["similarity"]
        assert g["det"]["n_points"] == e["det"]["n_points"] and g["det"]["icp"]["iters"] == e["det"]["icp"]["iters"]
        assert np.array_equal(_bits(g["pose"]), _bits(e["pose"]))
    det.close()


def test_context_outlives_its_detectors_in_either_destroy_order():
    """fl_context_destroy before fl_detector_destroy used to free the stream the detector still synchronises on."""
    c = api.Context(0)
    det = api.Detector(c, 2, [5, 8])
    det.add_class(synth.make_bank("obj", 4, 2, 2, 640, 480, seed=1))
    det.finalize(640, 480)
    c.close()                # deferred: the detector is still alive
    det.close()              # releases the context


def test_hard_candidate_cap_fails_only_the_cluttered_frame(ctx, oracle):
    """max_candidates < 0 asks for a hard cap: a frame that exceeds it reports FL_ERR_OVERFLOW in its own status and the other
    frame of the batch keeps its (oracle-equal) result -- one bad frame must not take a batch down."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=5, n_views=4, n_random=20)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=2, max_candidates=-256)
    blank_b, blank_d = np.zeros_like(sc["bgr"]), np.zeros_like(sc["depth"])          # no gradient, no depth return: no candidates at all
    got = det.recognize_batch([sc["bgr"], blank_b], [sc["depth"], blank_d], sc["K"], -100.0, 6, 0.0, -3.0e38)
    assert got[0]["status"] == L.FL_ERR_OVERFLOW and got[0]["found"] == 0
    e = oracle.recognition(blank_b, blank_d, sc["K"], [5, 8], sc["bank"], -100.0, 6, 0.0, -3.0e38)
    assert got[1]["status"] == 0 and got[1]["found"] == e["found"] and got[1]["n_matches"] == e["n_matches"]
    det.close()


def test_refine_matches_argument_checks_and_result(ctx, oracle):
    """fl_refine_matches = the second half of Recognition() for a match the caller chose: refining the detector's own best
    match must give fl_recognize_batch's pose; wrong frames / templates / call order are refused."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=9, n_views=4, n_random=6)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(sc["bank"])
    det.finalize(640, 480, max_batch=2)
    params = L.RecognitionParams(75.0, 7, 0.0, -3.0e38, L.FL_ICP_PARITY)
    one = np.zeros(1, api.MATCH_DTYPE)
    with pytest.raises(api.FealessError) as ex:
        det.refine_matches([0], one, sc["K"], params)                 # nothing matched yet
    assert ex.value.code == L.FL_ERR_STATE
    ref = det.recognize_batch([sc["bgr"]], [sc["depth"]], sc["K"], 75.0, 7, 0.0, -3.0e38)[0]
    lists = det.match_batch([sc["bgr"], sc["bgr"]], [sc["depth"], sc["depth"]], 75.0)
    best = lists[1][0][:1]
    res = det.refine_matches([1], best, sc["K"], params)
    assert res[0].found == 1 and np.array_equal(_bits(np.array(list(res[0].pose), np.float32)), _bits(ref["pose"].reshape(-1)))
    for frames, m in (([2], best), ([-1], best)):
        with pytest.raises(api.FealessError) as ex:
            det.refine_matches(frames, m, sc["K"], params)
        assert ex.value.code == L.FL_ERR_INVALID
    bad = best.copy()
    bad["template_id"] = sc["bank"].n_pyramids
    with pytest.raises(api.FealessError) as ex:
        det.refine_matches([0], bad, sc["K"], params)
    assert ex.value.code == L.FL_ERR_INVALID
    # only a batch submit leaves depth frames to refine on: a single-frame Detector::match, a match on quantised images or
    # a multi-hypothesis recognition in between must not let fl_refine_matches read an older batch's depth
    det.match(sc["bgr"], sc["depth"], 75.0)
    with pytest.raises(api.FealessError) as ex:
        det.refine_matches([0], best, sc["K"], params)
    assert ex.value.code == L.FL_ERR_STATE
    det.match_batch([sc["bgr"]], [sc["depth"]], 75.0)
    assert det.refine_matches([0], best, sc["K"], params)[0].found == 1
    det.recognize_topk(sc["bgr"], sc["depth"], sc["K"], 2, 75.0, 7, 0.0, -3.0e38)
    with pytest.raises(api.FealessError) as ex:
        det.refine_matches([0], best, sc["K"], params)
    assert ex.value.code == L.FL_ERR_STATE
    det.match_batch([sc["bgr"]], [sc["depth"]], 75.0)
    det.match_quantized(oracle.quantize_pyramid(sc["bgr"], sc["depth"], 2), 75.0)
    with pytest.raises(api.FealessError) as ex:
        det.refine_matches([0], best, sc["K"], params)
    assert ex.value.code == L.FL_ERR_STATE
    det.close()


def test_large_batch_jobs_dealt_longest_first_keep_their_results(ctx, oracle):
    """A batch above four frames per CU hands the ICP launch its jobs longest first (k_icp_count + k_icp_order: workgroup b
    runs job order[b]).  The order is scheduling only: every frame's result must be what the same frame gives alone, and what
    the batch gives in frame order (option icp_order = 0), bit for bit -- frames of different cloud sizes, so that the order is a
    real permutation.  Also the exact pruning of the scan (option scan_prune = 0 / 1) and a capped ICP occupancy (icp_wg_per_cu): same results either way."""
    import torch
    scenes = [synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=s, n_views=3) for s in (3, 5)]
    sc = scenes[0]
    frames_b, frames_d = [], []
    for k in range(6):                                       # six distinct frames: two scenes x three shifts (different crops -> cloud sizes)
        s2 = scenes[k % 2]
        sh = 14 * (k // 2)
        frames_b.append(np.roll(s2["bgr"], sh, axis=1))
        frames_d.append(np.roll(s2["depth"], sh, axis=1))
    bank = sc["bank"]
    n = 1100                                                 # > 4 x 256 CUs
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(bank)
    det.finalize(640, 480, max_batch=n, max_candidates=4096)
    d_b = torch.from_numpy(np.stack(frames_b)).cuda()
    d_d = torch.from_numpy(np.stack(frames_d).view(np.int16)).cuda()
    torch.cuda.synchronize()
    order = [(7 * i) % 6 for i in range(n)]
    bp = [d_b.data_ptr() + o * 640 * 480 * 3 for o in order]
    dp = [d_d.data_ptr() + o * 640 * 480 * 2 for o in order]
    params = L.RecognitionParams(75.0, 8, 0.0, -3.0e38, L.FL_ICP_PARITY)
    single = det.recognize_batch(frames_b, frames_d, sc["K"], 75.0, 8, 0.0, -3.0e38)
    assert sum(r["found"] for r in single) >= 3 and len({r["det"]["n_points"] for r in single if r["found"]}) >= 2
    runs = {}
    for tag, opts in (("longest_first", {}), ("frame_order", {"icp_order": 0}), ("unpruned", {"scan_prune": 0}),
                      ("three_workgroups_per_cu", {"icp_wg_per_cu": 3})):
        before = {k_: ctx.get_option(k_) for k_ in opts}
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        det.recognize_submit_device(bp, dp, sc["K"], params)
        runs[tag] = [api.recognition_result_to_dict(r) for r in det.recognize_collect(n)]
        for k_, v_ in before.items():
            ctx.set_option(k_, v_)
    for tag, res in runs.items():
        for i in range(n):
            e, g = single[order[i]], res[i]
            assert g["status"] == 0 and g["found"] == e["found"] and g["n_matches"] == e["n_matches"] and g["best"] == e["best"], (tag, i)
            if e["found"]:
                assert g["det"]["n_points"] == e["det"]["n_points"] and np.array_equal(_bits(g["pose"]), _bits(e["pose"])), (tag, i)
    det.close()
