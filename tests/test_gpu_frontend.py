"""GPU parity: quantisation front-end kernels vs the oracle, bit-exact (integer stages exactly,
float stages operator by operator; OpenCV itself is unavailable -> parity unpinned)."""
import numpy as np
import pytest

from fealess_amd import api, synth

pytestmark = pytest.mark.gpu


def _scene(seed, w=640, h=480):
    R, t = synth.object_pose(tx=20.0 * (seed % 3 - 1), ty=-10.0, tz=640.0 + 10 * seed, yaw=0.2 * seed)
    return synth.render(w, h, R, t, seed=seed)


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (320, 240, 2), (1280, 720, 3), (97, 61, 4), (33, 18, 5)])
def test_quantized_orientations_bit_exact(ctx, oracle, w, h, seed):
    if seed <= 3:
        _, bgr, _ = _scene(seed, w, h)
    else:
        bgr = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = ctx.quantized_orientations(bgr, 10.0)
    exp = oracle.quantized_orientations(bgr, 10.0)
    assert np.array_equal(got, exp), int((got != exp).sum())
    if seed <= 3:
        assert (exp != 0).sum() > 500          # not vacuous


def test_quantized_orientations_noise_image(ctx, oracle):
    bgr = np.random.default_rng(9).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    assert np.array_equal(ctx.quantized_orientations(bgr, 10.0), oracle.quantized_orientations(bgr, 10.0))


@pytest.mark.parametrize("w,h,seed", [(640, 480, 1), (1280, 720, 2), (64, 48, 3), (23, 17, 4)])
def test_quantized_normals_bit_exact(ctx, oracle, w, h, seed):
    if seed <= 2:
        depth, _, _ = _scene(seed, w, h)
    else:
        depth = (600 + np.random.default_rng(seed).integers(0, 60, (h, w))).astype(np.uint16)
    depth = depth.copy()
    depth[5:9, 7:15] = 0                       # sensor holes
    got = ctx.quantized_normals(depth)
    exp = oracle.quantized_normals(depth)
    assert np.array_equal(got, exp), int((got != exp).sum())
    if seed <= 2:
        assert (exp != 0).sum() > 1000


def test_normals_far_and_thresholds(ctx, oracle):
    depth = np.full((120, 160), 2500, np.uint16)               # beyond distance_threshold -> all zero
    assert not ctx.quantized_normals(depth).any()
    d2, _, _ = _scene(2, 320, 240)
    assert np.array_equal(ctx.quantized_normals(d2, 900, 20), oracle.quantized_normals(d2, 900, 20))


@pytest.mark.parametrize("dist_thr,diff_thr", [(2000, 400), (2000, 401), (65535, 5000), (65535, 65535), (2000, 1), (2000, 0), (2000, -3)])
def test_normals_difference_threshold_paths(ctx, oracle, dist_thr, diff_thr):
    """The kernel accumulates in int32 for difference_threshold <= 400 and widens the last products beyond it:
    both paths, their boundary and the degenerate gates, on smooth and on full-range random depth."""
    d, _, _ = _scene(3, 320, 240)
    rough = np.random.default_rng(11).integers(0, 65536, (96, 128)).astype(np.uint16)
    steps = (500 + 300 * (np.random.default_rng(12).integers(0, 4, (96, 128)))).astype(np.uint16)
    for depth in (d, rough, steps):
        got, exp = ctx.quantized_normals(depth, dist_thr, diff_thr), oracle.quantized_normals(depth, dist_thr, diff_thr)
        assert np.array_equal(got, exp), int((got != exp).sum())


@pytest.mark.parametrize("w,h", [(640, 480), (320, 240), (1280, 720), (321, 241), (50, 34)])
def test_pyrdown_bit_exact(ctx, oracle, w, h):
    bgr = np.random.default_rng(w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(ctx.pyrdown_bgr(bgr), oracle.pyrdown_bgr(bgr))


@pytest.mark.parametrize("levels,T", [(2, [5, 8]), (1, [5])])
def test_match_from_images_bit_exact(ctx, oracle, levels, T):
    """Detector::match from BGR + depth: quantized pyramid and final match list equal the oracle's."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=levels, seed=5, n_views=4,
                                 n_random=20)
    det = api.Detector(ctx, 2, T)
    det.add_class(sc["bank"])
    det.finalize(640, 480)
    got, n_got = det.match(sc["bgr"], sc["depth"], 70.0)
    exp, n_exp = oracle.match_images(sc["bgr"], sc["depth"], T, [sc["bank"]], 70.0)
    for a, b in zip(det.last_quantized(), oracle.quantize_pyramid(sc["bgr"], sc["depth"], levels)):
        assert np.array_equal(a, b)
    assert n_exp > 0 and n_got == n_exp
    for k in ("x", "y", "class_idx", "template_id"):
        assert np.array_equal(got[k], exp[k])
    assert np.array_equal(got["similarity"].view(np.uint32), exp["similarity"].view(np.uint32))
    det.close()


@pytest.mark.parametrize("which", ["both", "color_only", "depth_only", "empty"])
def test_match_with_masks_bit_exact(ctx, oracle, which):
    """Detector::match's masks argument (linemod.cpp:445-459, 733-745, 1364-1379)."""
    T, thr = [5, 8], 30.0
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=5, n_views=4, n_random=20)
    rng = np.random.default_rng(11)
    mc = (rng.random((480, 640)) < 0.5).astype(np.uint8) * 255       # speckled: exercises the NN mask pyramid
    md = np.zeros((480, 640), np.uint8)
    md[40:440, 101:345] = 7                                          # any non-zero value counts; cuts the object
    masks = {"both": [mc, md], "color_only": [mc, None], "depth_only": [None, md], "empty": [None, None]}[which]
    det = api.Detector(ctx, 2, T)
    det.add_class(sc["bank"])
    det.finalize(640, 480)
    got, n_got = det.match(sc["bgr"], sc["depth"], thr, masks=masks)
    exp, n_exp = oracle.match_images(sc["bgr"], sc["depth"], T, [sc["bank"]], thr, masks=masks)
    plain, n_plain = oracle.match_images(sc["bgr"], sc["depth"], T, [sc["bank"]], thr)
    assert n_got == n_exp and n_exp > 0
    if which == "empty":
        assert np.array_equal(exp, plain)
    else:
        assert not np.array_equal(exp[:1], plain[:1])                # the mask changed the outcome
    for k in ("x", "y", "class_idx", "template_id"):
        assert np.array_equal(got[k], exp[k])
    assert np.array_equal(got["similarity"].view(np.uint32), exp["similarity"].view(np.uint32))
    q = det.last_quantized()
    if which in ("both", "depth_only"):
        assert not q[1][:40].any() and not q[3][:20].any() and q[1][40:440, 101:345].any()
    with pytest.raises(api.FealessError):
        det.match(sc["bgr"], sc["depth"], thr, masks=[mc])           # masks.size() != modalities.size()
    det.close()


@pytest.mark.parametrize("sw,sh,dw,dh", [(1280, 960, 640, 480), (800, 600, 640, 480), (320, 240, 640, 480), (1024, 768, 640, 480),
                                         (641, 481, 640, 480), (1280, 720, 640, 360), (37, 23, 64, 48)])
def test_resize_linear_bit_exact(ctx, oracle, sw, sh, dw, dh):
    """PrepareInputData's cv::resize(INTER_LINEAR) (obj_reco_lmicp.cpp:39-45, 248-249) for BGR8 and depth16."""
    rng = np.random.default_rng(sw)
    bgr = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    depth = rng.integers(0, 65536, (sh, sw)).astype(np.uint16)
    assert np.array_equal(ctx.resize_linear(bgr, dw, dh), oracle.resize_linear_u8(bgr, dw, dh))
    assert np.array_equal(ctx.resize_linear(depth, dw, dh), oracle.resize_linear_u16(depth, dw, dh))


def test_c3_match_from_images_1280x720_three_levels(ctx, oracle):
    """BASELINE configs[2] geometry end to end: Detector::match from BGR + depth at 1280x720 with T = {5, 8, 4}
    (linemod.cpp:1356-1441), single-frame entry and the batched one, against orc_match_images."""
    from fealess_amd.bank import TemplateBank
    w, h, T = 1280, 720, [5, 8, 4]
    K = (915.0, 915.0, 640.0, 360.0)
    rng = np.random.default_rng(3)
    R, t = synth.object_pose(tx=30.0, ty=-20.0, tz=660.0)
    frames = []
    bank = TemplateBank("obj", 3, 2)
    for s in range(2):
        Rs = synth.rot_z(0.05 * s) @ R
        depth, bgr, _ = synth.render(w, h, Rs, t + np.array([25.0 * s, 0, 0]), seed=40 + s, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
        frames.append((bgr, depth))
        for v in range(2):
            dR = synth.rot_z(np.deg2rad(rng.uniform(-2, 2))) @ synth.rot_x(np.deg2rad(rng.uniform(-2, 2)))
            tv = t + np.array([25.0 * s, 0, 0]) + rng.uniform(-10, 10, 3)
            d_bg, bgr_v, mask = synth.render(w, h, dR @ Rs, tv, seed=60 + 2 * s + v, noise=False, fx=K[0], fy=K[1], cx=K[2], cy=K[3])
            ex = ctx.extract_template_pyramid(bgr_v, d_bg, (mask * 255).astype(np.uint8), 3)
            assert ex is not None
            bank.add_pyramid(ex[0], synth.pose13(dR @ Rs, tv), None)
    while bank.n_pyramids < 240:                                  # 4 trained views + 236 random pyramids
        bank.add_pyramid(synth.random_pyramid(rng, 3, 2, w, h), None, None)
    det = api.Detector(ctx, 2, T)
    det.add_class(bank)
    det.finalize(w, h, max_batch=2)
    thr = 65.0
    exps = [oracle.match_images(b, d, T, [bank], thr) for b, d in frames]
    assert all(n > 0 for _, n in exps)
    for (b, d), (exp, n_exp) in zip(frames, exps):
        got, n_got = det.match(b, d, thr)
        for a, e in zip(det.last_quantized(), oracle.quantize_pyramid(b, d, 3)):
            assert np.array_equal(a, e)
        assert n_got == n_exp and got.tobytes() == exp[:len(got)].tobytes()
    # the batched entry (lazy fine levels) gives the same lists
    both = det.match_batch([f[0] for f in frames], [f[1] for f in frames], thr)
    for (got, n_got), (exp, n_exp) in zip(both, exps):
        assert n_got == n_exp and got.tobytes() == exp[:len(got)].tobytes()
    # fl_similarity_maps is a debug tap: the match lists of the batch survive it
    det.similarity_maps(0, 4)
    again, n_again = det.match_batch_collect(0)
    assert n_again == exps[0][1] and again.tobytes() == exps[0][0][:len(again)].tobytes()
    det.close()
