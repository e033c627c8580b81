"""CPU tests of the ORACLE (oracle/liboracle.so): tables vs the reference text (when present),
independent naive re-derivations on tiny inputs, hand-worked known answers, domain properties,
and the committed golden fixtures.  No GPU needed.

The reference holds no tests or golden vectors for this path, so nothing
here can pin the oracle to the reference's *outputs*: "parity unpinned".  What is pinned: the two
data tables against the reference's own text, the algorithm against independent formulations,
and the oracle against its committed fixtures.
"""
import os
import re

import numpy as np
import pytest

import util
from fealess_amd import synth
from fealess_amd.bank import TemplateBank

REF = "/root/reference"


# ---- tables ------------------------------------------------------------------------------------
def test_similarity_lut_rule(oracle):
    lut = oracle.similarity_lut()
    for ori in range(8):
        for byte in range(256):
            a = lut[32 * ori + (byte & 15)]
            b = lut[32 * ori + 16 + (byte >> 4)]
            assert max(a, b) == util.naive_response(byte, ori)
    # golden digest so the table cannot drift silently where the reference is absent
    assert int(lut.astype(np.int64).sum()) == 460 and int((lut.astype(np.int64) * np.arange(256)).sum()) == 59268


@pytest.mark.skipif(not os.path.exists(REF), reason="reference text not available on this box")
def test_tables_equal_reference_text(oracle):
    src = open(os.path.join(REF, "linemod", "linemod.cpp"), errors="replace").read()
    line = [l for l in src.splitlines() if l.startswith("CV_DECL_ALIGNED(16) static const unsigned char SIMILARITY_LUT")][0]
    ref_lut = np.array([int(v) for v in re.findall(r"\d+", line[line.index("{"):])], np.uint8)
    assert ref_lut.size == 256 and np.array_equal(ref_lut, oracle.similarity_lut())
    txt = open(os.path.join(REF, "linemod", "normal_lut.i")).read()
    nums = [int(v) for v in re.findall(r"\d+", txt[txt.index("{"):])]
    assert np.array_equal(np.array(nums[:8000], np.uint8), oracle.normal_lut())


def test_normal_lut_digest(oracle):
    lut = oracle.normal_lut().reshape(20, 20, 20)
    assert all(np.array_equal(lut[z], lut[0]) for z in range(20))
    vals, counts = np.unique(lut, return_counts=True)
    assert dict(zip(vals.tolist(), counts.tolist())) == {1: 760, 2: 1060, 4: 740, 8: 1160, 16: 920, 32: 1280, 64: 920, 128: 1160}


# ---- scan stages vs naive formulations ---------------------------------------------------------
@pytest.mark.parametrize("T", [2, 4, 5])
def test_spread_and_response_vs_naive(oracle, T):
    rng = np.random.default_rng(T)
    q = synth.random_quantized(rng, 20 * T // T * 4, 12, 0.3)[:12, :16]
    q = np.ascontiguousarray(q)
    sp = oracle.spread(q, T)
    assert np.array_equal(sp, util.naive_spread(q, T))
    maps = oracle.response_maps(sp)
    for ori in range(8):
        exp = np.vectorize(lambda b: util.naive_response(int(b), ori))(sp)
        assert np.array_equal(maps[ori], exp)


def test_linear_memory_layout(oracle):
    rng = np.random.default_rng(3)
    T, w, h = 4, 32, 16
    q = synth.random_quantized(rng, w, h, 0.3)
    lm = oracle.build_linear_memories(q, T)
    sp = oracle.spread(q, T)
    maps = oracle.response_maps(sp)
    W, H = w // T, h // T
    for lab in (0, 5):
        for y in range(h):
            for x in range(w):
                assert lm[lab, ((y % T) * T + x % T) * W * H + (y // T) * W + x // T] == maps[lab, y, x]
    assert not lm[:, T * T * W * H:].any()          # zero pad


def test_similarity_vs_naive_including_wrap_and_overread(oracle):
    rng = np.random.default_rng(8)
    T, w, h = 4, 48, 32
    q = synth.random_quantized(rng, w, h, 0.4)
    lm = oracle.build_linear_memories(q, T)
    for (width, height, feats) in [
        (12, 8, [(0, 0, 1), (5, 3, 6), (11, 7, 2), (3, 8, 4)]),       # y == height == 8, 8 % 4 == 0: over-read (Q2)
        (20, 13, [(19, 12, 0), (7, 2, 3), (-1, 2, 3), (60, 1, 2)]),    # out-of-image features skipped
        (47, 31, [(0, 0, 7)]),                                         # a single template position
    ]:
        b = TemplateBank("o", 1, 1)
        b.add_pyramid([dict(width=width, height=height, offset_x=0, offset_y=0, pyramid_level=0,
                            features=np.array(feats, np.int32))])
        got = oracle.total_similarity([lm], b, 0, w, h, T)
        assert np.array_equal(got, util.naive_similarity(q, T, feats, width, height))


def test_match_known_answer_single_level(oracle):
    """Hand-worked: one orientation-0 pixel at (x, y) = (12, 9); a one-feature template (label 0,
    feature at its origin) of size 4x4; T = 4.  spread (anchored top-left) sets bit 0 on
    x in 9..12, y in 6..9; response 4 there.  Template position (c, r) samples (4c, 4r): only
    (12, 8) = (c, r) = (3, 2) lies inside.  raw threshold for 80 %: (int)(2 + 0.8*2 + 0.5) = 4 and
    4 > 4 is false: no match; at 70 %: (int)(2 + 1.4 + 0.5) = 3 -> one match at
    (c*T + T/2 - 1, r*T + T/2 - 1) = (13, 9) with similarity 4*100/4 + 0.5 = 100.5 (Q3)."""
    q = np.zeros((16, 24), np.uint8)
    q[9, 12] = 1
    b = TemplateBank("o", 1, 1)
    b.add_pyramid([dict(width=4, height=4, offset_x=0, offset_y=0, pyramid_level=0, features=np.array([[0, 0, 0]], np.int32))])
    m, n = oracle.match_quantized([q], 24, 16, [4], [b], 80.0)
    assert n == 0
    m, n = oracle.match_quantized([q], 24, 16, [4], [b], 70.0)
    assert n == 1
    assert (int(m[0]["x"]), int(m[0]["y"])) == (13, 9)
    assert m[0]["similarity"] == np.float32(100.5)
    sim = oracle.total_similarity([oracle.build_linear_memories(q, 4)], b, 0, 24, 16, 4)
    assert sim[2, 3] == 4 and sim.sum() == 4


def test_sort_unique_semantics(oracle):
    """Match::operator< orders by similarity desc then template_id asc; operator== ignores the
    template id (linemod.hpp:262-274): equal (x, y, sim) from adjacent templates collapse."""
    rng = np.random.default_rng(5)
    qs = [synth.random_quantized(rng, 160 >> l, 128 >> l, 0.05) for l in range(2)]
    bank = synth.make_bank("o", 6, 2, 1, 160, 128, seed=2, qs=qs, planted_frac=1.0, bbox=48)
    t, f, p = bank.arrays()
    dup = TemplateBank("o", 2, 1)          # every pyramid twice -> identical matches, different ids
    for i in range(6):
        for _ in range(2):
            tl = []
            for k in range(2):
                hdr = t[i * 2 + k]
                fr = f[hdr["feat_begin"]:hdr["feat_begin"] + hdr["feat_count"]]
                tl.append(dict(width=int(hdr["width"]), height=int(hdr["height"]), offset_x=int(hdr["offset_x"]),
                               offset_y=int(hdr["offset_y"]), pyramid_level=k, features=np.stack([fr["x"], fr["y"], fr["label"]], 1)))
            dup.add_pyramid(tl)
    m1, n1 = oracle.match_quantized(qs, 160, 128, [4, 8], [bank], 80.0)
    m2, n2 = oracle.match_quantized(qs, 160, 128, [4, 8], [dup], 80.0)
    assert n1 > 0
    sim = m2["similarity"]
    assert np.all(sim[:-1] >= sim[1:])
    key = lambda m: sorted({(int(a["x"]), int(a["y"]), float(a["similarity"])) for a in m})
    assert key(m1) == key(m2)
    same = (m2["x"][1:] == m2["x"][:-1]) & (m2["y"][1:] == m2["y"][:-1]) & (sim[1:] == sim[:-1])
    assert not same.any()                  # adjacent duplicates removed


# ---- front-end properties ----------------------------------------------------------------------
def test_fast_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        x, y = rng.integers(-1020, 1021, 2)
        if x == 0 and y == 0:
            continue
        a = oracle.lib().orc_fast_atan2(float(y), float(x))
        t = np.degrees(np.arctan2(y, x)) % 360.0
        assert abs(((a - t + 180) % 360) - 180) < 0.35
    assert oracle.lib().orc_fast_atan2(0.0, 0.0) == 0.0


def test_orientation_of_a_vertical_edge(oracle):
    bgr = np.zeros((40, 60, 3), np.uint8)
    bgr[:, 30:] = 200                                   # dx > 0 -> 0 deg -> bin 0 -> bit 1
    q = oracle.quantized_orientations(bgr, 10.0)
    assert set(np.unique(q)) == {0, 1}
    assert q[5:35, 27:33].any() and not q[:, :20].any()
    bgr2 = np.ascontiguousarray(bgr.transpose(1, 0, 2))  # horizontal edge: 90 deg -> bin 4 -> bit 16
    q2 = oracle.quantized_orientations(bgr2, 10.0)
    vals, counts = np.unique(q2, return_counts=True)
    assert set(vals) <= {0, 1, 16} and dict(zip(vals.tolist(), counts.tolist()))[16] >= 250


def test_normals_of_planes(oracle):
    d = np.full((60, 80), 800, np.uint16)
    q = oracle.quantized_normals(d)
    inner = q[8:50, 8:70]
    assert len(np.unique(inner)) == 1 and inner[0, 0] != 0   # fronto-parallel: one label everywhere
    assert not q[:5].any() and not q[:, :5].any()           # untouched border stays 0
    ys = np.arange(60)[:, None].repeat(80, 1)
    up = oracle.quantized_normals((800 + 2 * ys).astype(np.uint16))[8:50, 8:70]
    down = oracle.quantized_normals((800 - 2 * ys).astype(np.uint16))[8:50, 8:70]
    assert np.all(up == 4) and np.all(down == 64)          # +y / -y tilt: directions 90 and 270 degrees
    assert not oracle.quantized_normals(np.full((60, 80), 2100, np.uint16)).any()


def test_pyrdown_and_median_properties(oracle):
    c = np.full((24, 36, 3), 77, np.uint8)
    assert np.all(oracle.pyrdown_bgr(c) == 77)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (24, 36, 3), dtype=np.uint8)
    pd = oracle.pyrdown_bgr(img)
    assert pd.shape == (12, 18, 3)
    k = np.array([1, 4, 6, 4, 1])
    acc = 0
    for j in range(5):
        for i in range(5):
            acc += k[j] * k[i] * int(img[2 * 5 + j - 2, 2 * 7 + i - 2, 1])
    assert pd[5, 7, 1] == (acc + 128) >> 8


# ---- ICP / back-projection ---------------------------------------------------------------------
def test_depth_to_3d_formula(oracle):
    d = np.array([[0, 1000], [500, 2000]], np.uint16)
    p = oracle.depth_to_3d(d, 600.0, 500.0, 1.0, 0.5)
    assert np.isnan(p[0, 0]).all()
    z = np.float32(1000) * np.float32(0.001)
    assert p[0, 1, 2] == z and p[0, 1, 0] == (np.float32(1) - np.float32(1.0)) * (np.float32(1) / np.float32(600)) * z
    assert p[1, 0, 1] == ((np.float32(1) - np.float32(0.5)) * (np.float32(1) / np.float32(500))) * (np.float32(500) * np.float32(0.001))


def test_svd3_reconstructs(oracle):
    rng = np.random.default_rng(4)
    for _ in range(20):
        A = rng.normal(0, 100, (3, 3)).astype(np.float32)
        w, u, vt = oracle.svd3(A)
        assert np.all(w[:-1] >= w[1:])
        assert np.allclose(u @ np.diag(w) @ vt, A, atol=2e-3 * np.abs(A).max())
        assert np.allclose(u.T @ u, np.eye(3), atol=1e-4) and np.allclose(vt @ vt.T, np.eye(3), atol=1e-4)


def _paired_clouds(seed, n=1200):
    rng = np.random.default_rng(seed)
    ref = (rng.normal(0, 1, (n, 3)) * np.array([60, 40, 12]) + np.array([10, -5, 650])).astype(np.float32)
    dR = synth.rot_z(0.03) @ synth.rot_x(-0.02)
    c = ref.mean(0)
    model = ((ref - c) @ dR.T + c + np.array([2.0, -1.0, 1.5])).astype(np.float32)
    return ref, model, dR


def test_icp_recovers_rigid_motion_and_kdtree_equals_brute(oracle):
    ref, model, dR = _paired_clouds(1)
    a = oracle.icp(ref, model, 15, 0.0, -3.0e38, use_kdtree=True)
    b = oracle.icp(ref, model, 15, 0.0, -3.0e38, use_kdtree=False)
    assert a["iters"] == 15
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["T"], b["T"])
    moved = model @ a["R"].T + a["T"]
    before = np.linalg.norm(model - ref, axis=1).mean()
    after = np.linalg.norm(moved - ref, axis=1).mean()
    assert after < 0.5 * before and abs(float(a["dist_mean"]) - after) < 0.5
    assert np.abs(a["R"] - dR.T).max() < 5e-2
    c = oracle.icp(ref, model, 15, 0.0, -3.0e38, accum64=True)
    assert np.abs(c["R"] - a["R"]).max() < 5e-3       # float32 summation noise of the reference itself


def test_icp_reference_quirks(oracle):
    ref, model, _ = _paired_clouds(2, 300)
    r = oracle.icp(ref[:2], model[:2], 5)
    assert r["rc"] == -1 and r["dist_mean"] == -1.0 and not r["R"].any()        # < 3 points (ICP.cpp:633-638)
    r = oracle.icp(ref, model, 10, 0.5, 0.01)                                   # CadReco defaults: early exit
    assert 1 <= r["iters"] <= 10
    far = model.copy()
    far[:, 0] += 400.0
    r = oracle.icp(ref, far, 6, 0.0, -3.0e38)
    assert r["iters"] == 6


# ---- golden fixtures: the oracle must keep producing what was committed ---------------------------
def test_golden_linemod(oracle):
    g = util.golden("linemod_320x160.npz")
    bank = util.bank_from_arrays(g["templates"], g["features"], g["poses"], 2, 2)
    qs = [g["q0"], g["q1"], g["q2"], g["q3"]]
    m, n = oracle.match_quantized(qs, 320, 160, [5, 8], [bank], float(g["threshold"]))
    assert n == int(g["n_matches"]) and m.tobytes() == g["matches"].tobytes()
    lms = [oracle.build_linear_memories(qs[2 + k], 8) for k in range(2)]
    for i in range(bank.n_pyramids):
        assert np.array_equal(oracle.total_similarity(lms, bank, i, 160, 80, 8), g["sims"][i])


def test_golden_frontend_icp_recognition(oracle):
    g = util.golden("frontend_256x192.npz")
    assert np.array_equal(oracle.quantized_orientations(g["bgr"], 10.0), g["qo"])
    assert np.array_equal(oracle.quantized_normals(g["depth"]), g["qn"])
    assert np.array_equal(oracle.pyrdown_bgr(g["bgr"]), g["pyrdown"])
    g = util.golden("icp_1500.npz")
    r = oracle.icp(g["ref"], g["model"], 12, 0.0, -3.0e38)
    assert np.array_equal(r["R"], g["R32"]) and np.array_equal(r["T"], g["T32"])
    g = util.golden("recognition_vga.npz")
    bank = util.bank_from_arrays(g["templates"], g["features"], g["poses"], 2, 2, model_depths=g["model_depths"])
    r = oracle.recognition(g["bgr"], g["depth"], tuple(g["K"]), [5, 8], bank, 75.0, 10, 0.5, 0.01)
    assert r["found"] == 1 and np.array_equal(r["pose"], g["pose"])
    assert [r["best"]["x"], r["best"]["y"], r["best"]["template_id"]] == g["best"].tolist()


def test_masked_match_semantics(oracle):
    """masks (linemod.cpp:445-459, 733-745): equal to matching the pyramid whose level-l image is zeroed where the
    l-times NN-halved mask is zero; all-ones masks change nothing; an all-zero mask removes every match."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=5, n_views=3,
                                 n_random=5, w=320, h=240)
    T = [5, 8]
    bank = [sc["bank"]]
    plain, n_plain = oracle.match_images(sc["bgr"], sc["depth"], T, bank, 60.0)
    ones = np.ones((240, 320), np.uint8)
    same, n_same = oracle.match_images(sc["bgr"], sc["depth"], T, bank, 60.0, masks=[ones, ones])
    assert n_plain > 0 and n_same == n_plain and np.array_equal(same, plain)
    none, n_none = oracle.match_images(sc["bgr"], sc["depth"], T, bank, 60.0, masks=[ones * 0, ones * 0])
    assert n_none == 0
    rng = np.random.default_rng(3)
    mc = (rng.random((240, 320)) < 0.8).astype(np.uint8)
    md = (rng.random((240, 320)) < 0.8).astype(np.uint8) * 200
    got, n_got = oracle.match_images(sc["bgr"], sc["depth"], T, bank, 60.0, masks=[mc, md])
    q = oracle.quantize_pyramid(sc["bgr"], sc["depth"], 2)
    q[0] = q[0] * (mc != 0)
    q[1] = q[1] * (md != 0)
    q[2] = q[2] * (mc[::2, ::2] != 0)
    q[3] = q[3] * (md[::2, ::2] != 0)
    exp, n_exp = oracle.match_quantized(q, 320, 240, T, bank, 60.0)
    assert n_got == n_exp and np.array_equal(got, exp)


def test_resize_linear_properties(oracle):
    """cv::resize(INTER_LINEAR) restatement (obj_reco_lmicp.cpp:39-45): exact 2x2 decimation is the rounded box mean
    (OpenCV redirects it to INTER_AREA), constants are preserved, a ramp stays the ramp it samples."""
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    box = ((a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear_u8(a, 64, 48), box)
    d = rng.integers(0, 4000, (96, 128)).astype(np.uint16)
    boxd = ((d[0::2, 0::2].astype(int) + d[0::2, 1::2] + d[1::2, 0::2] + d[1::2, 1::2] + 2) >> 2).astype(np.uint16)
    assert np.array_equal(oracle.resize_linear_u16(d, 64, 48), boxd)
    assert (oracle.resize_linear_u8(np.full((30, 40, 3), 77, np.uint8), 64, 48) == 77).all()
    assert (oracle.resize_linear_u16(np.full((30, 40), 1234, np.uint16), 64, 48) == 1234).all()
    ramp = (np.arange(80)[None, :] * 10 + np.zeros((60, 1))).astype(np.uint16)
    r = oracle.resize_linear_u16(ramp, 64, 48)
    x = (np.arange(64) + 0.5) * 1.25 - 0.5                       # source coordinate of every output column
    assert np.array_equal(r[7, 1:-1], np.rint(x[1:-1] * 10).astype(np.uint16)) and (r == r[0]).all()
    r8 = oracle.resize_linear_u8((np.arange(80)[None, :, None] * 3 + np.zeros((60, 1, 3))).astype(np.uint8), 64, 48)
    assert np.abs(r8[5, 1:-1, 0].astype(float) - x[1:-1] * 3).max() < 1.0   # 11-bit taps, truncating shifts
    up = oracle.resize_linear_u8(a, 256, 192)                     # enlarging: borders replicate (sx < 0 -> 0, fx = 0)
    assert np.array_equal(up[0, 0], a[0, 0]) and np.array_equal(up[-1, -1], a[-1, -1])


def test_erode_and_distance_transform_pinned_against_scipy(oracle):
    """The two OpenCV calls of extractTemplate, restated in oracle/extract_oracle.c, against an independent
    implementation: cv::erode (3x3 rectangle, n iterations, BORDER_REPLICATE) = (2n+1)^2 minimum filter with edge
    replication; cv::distanceTransform(DIST_C, 3) = exact chessboard distance to the nearest zero pixel."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(0)
    for shape, p in (((60, 80), 0.8), ((33, 47), 0.95), ((5, 7), 0.5)):
        m = (rng.random(shape) < p).astype(np.uint8) * 255
        for it in (1, 2):
            assert np.array_equal(oracle.erode_rect(m, it), ndi.minimum_filter(m, size=2 * it + 1, mode="nearest"))
        img = (rng.random(shape) < 0.97).astype(np.uint8) * 64
        if img.all():
            img[0, 0] = 0
        assert np.array_equal(oracle.distance_transform_c3(img),
                              ndi.distance_transform_cdt(img != 0, metric="chessboard").astype(np.float32))
    assert (oracle.distance_transform_c3(np.full((5, 7), 3, np.uint8)) == 8192.0).all()   # no zero pixel: capped INIT_DIST0


def test_extract_template_semantics(oracle):
    """extractTemplate / selectScatteredFeatures / cropTemplates / addTemplate (linemod.cpp:52-164, 461-513, 747-825,
    1579-1615): hand-checkable properties of the restatement."""
    R, t = synth.object_pose(tz=650.0)
    depth, bgr, mask = synth.render(640, 480, R, t, seed=3, noise=False, background=True)
    m255 = (mask * 255).astype(np.uint8)
    out = oracle.add_template(bgr, depth, m255, 2)
    assert out is not None
    tl, feats, bb = out
    assert [len(f) for f in feats] == [63, 63, 31, 31] and bb[0] % 2 == 0 and bb[1] % 2 == 0
    for k in range(4):
        l = int(tl[k]["pyramid_level"])
        assert l == k // 2 and tl[k]["width"] == bb[2] >> l and tl[k]["offset_x"] == bb[0] >> l
        f = feats[k]
        assert f["x"].min() >= 0 and f["y"].min() >= 0 and f["x"].max() <= tl[k]["width"] and (0 <= f["label"]).all() and (f["label"] < 8).all()
        assert len({(int(a), int(b)) for a, b in zip(f["x"], f["y"])}) == len(f)        # scattered: no position twice
    # colour features lie on the border of the mask (mask - erode(mask)), depth features inside the 5x5-eroded mask
    q, mag = oracle.quantized_orientations_mag(bgr)
    border = (m255 > 0) & (oracle.erode_rect(m255, 1) == 0)
    f = feats[0]
    ax, ay = f["x"] + tl[0]["offset_x"], f["y"] + tl[0]["offset_y"]
    assert border[ay, ax].all() and (mag[ay, ax] > 55.0 ** 2).all() and np.array_equal(1 << f["label"], q[ay, ax])
    f = feats[1]
    ax, ay = f["x"] + tl[1]["offset_x"], f["y"] + tl[1]["offset_y"]
    qn = oracle.quantized_normals(depth)
    assert (oracle.erode_rect(m255, 2)[ay, ax] > 0).all() and np.array_equal(1 << f["label"], qn[ay, ax])
    # the stand-alone entry point with a candidate set of exactly num_features returns them all (any order of choice)
    qq = np.zeros((40, 40), np.uint8)
    mg = np.zeros((40, 40), np.float32)
    pts = [(3, 4), (20, 7), (35, 30), (9, 33)]
    for i, (x, y) in enumerate(pts):
        qq[y, x] = 1 << i
        mg[y, x] = 4000.0 + i
    got = oracle.extract_template_color(qq, mg, None, 55.0, 4)
    assert got[0]["x"] == 9 and got[0]["label"] == 3                                  # highest score first
    assert sorted((int(a), int(b)) for a, b in zip(got["x"], got["y"])) == sorted(pts)
    assert oracle.extract_template_color(qq, mg, None, 55.0, 5) is None                # too few candidates
    assert oracle.add_template(np.zeros((480, 640, 3), np.uint8), np.full((480, 640), 1000, np.uint16), None, 2) is None


def test_nms_semantics(oracle):
    """nonMaximumSuppression (ICP/NMS.cpp:6-40) on hand-made hypotheses: groups by distance to the group's CURRENT
    best, replacement needs > 0.85 x the opener's points and a smaller ICP distance."""
    import ctypes as C
    objs = (oracle.OrcRecognitionResult * 5)()

    def setup(i, t, npts, dist):
        objs[i].det.T_final[0], objs[i].det.T_final[1], objs[i].det.T_final[2] = t
        objs[i].det.n_points = npts
        objs[i].det.icp.dist_mean = dist
    setup(0, (0, 0, 0), 1000, 2.0)
    setup(1, (5, 0, 0), 900, 1.0)        # near 0, enough points, better -> becomes the best of group 0
    setup(2, (12, 0, 0), 2000, 0.5)      # 12 from object 0 but 7 from the new best (object 1) -> absorbed; better -> wins
    setup(3, (100, 0, 0), 10, 0.1)       # far away -> its own group
    setup(4, (103, 0, 0), 8, 0.01)       # near 3 but 8 <= int(10 * 0.85) = 8 points -> absorbed without replacing
    win = (C.c_int * 5)()
    n = oracle.lib().orc_nms(objs, 5, C.c_float(10.0), win)
    assert [win[i] for i in range(n)] == [2, 3]
    n = oracle.lib().orc_nms(objs, 5, C.c_float(6.0), win)      # 2 is now 7 > 6 from the best of group 0
    assert [win[i] for i in range(n)] == [1, 2, 3]


def test_filters_and_nn_pinned_against_scipy(oracle):
    """The oracle's restatements of OpenCV's separable filters and of the exact nearest-neighbour search against
    independent implementations (scipy): what remains unpinned is only OpenCV's choice of kernel and rounding, as
    published -- not our arithmetic."""
    ndi = pytest.importorskip("scipy.ndimage")
    sp = pytest.importorskip("scipy.spatial")
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    # GaussianBlur 7x7, sigma from ksize, 8-bit path: separable {8,28,56,72,56,28,8}/256 twice, ONE rounding (acc + 2^15) >> 16
    k = np.array([8, 28, 56, 72, 56, 28, 8], np.int64)
    acc = ndi.correlate1d(ndi.correlate1d(img.astype(np.int64), k, axis=1, mode="nearest"), k, axis=0, mode="nearest")
    assert np.array_equal(oracle.gaussian7_bgr(img), ((acc + (1 << 15)) >> 16).astype(np.uint8))
    # medianBlur 5, BORDER_REPLICATE
    q = (1 << rng.integers(0, 8, (41, 29))).astype(np.uint8) * (rng.random((41, 29)) < 0.7)
    assert np.array_equal(oracle.median5(q.astype(np.uint8)), ndi.median_filter(q.astype(np.uint8), size=5, mode="nearest"))
    # pyrDown: [1 4 6 4 1]^2 / 256 with REFLECT_101 (scipy: "mirror"), (acc + 128) >> 8, even samples
    k5 = np.array([1, 4, 6, 4, 1], np.int64)
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    acc = ndi.correlate1d(ndi.correlate1d(big.astype(np.int64), k5, axis=1, mode="mirror"), k5, axis=0, mode="mirror")
    assert np.array_equal(oracle.pyrdown_bgr(big), ((acc[::2, ::2] + 128) >> 8).astype(np.uint8))
    # exact 1-NN of the ICP (kd-tree oracle path) against scipy's cKDTree: the number of pairs kept in iteration 2
    # (d^2 <= 3 * dist_mean, ICP.cpp:268) must equal the count obtained from an independent exact NN search on the
    # model transformed by iteration 1's (R, T)
    ref = rng.normal(0, 30, (700, 3)).astype(np.float32)
    ref[:, 2] += 400
    model = (ref + rng.normal(0, 0.7, ref.shape)).astype(np.float32)
    r = oracle.icp(ref, model, 2, 0.0, -3.0e38, accum64=False, use_kdtree=True, trace=True)
    assert r["iters"] == 2
    n1, dm1 = r["trace"][0][0], np.float32(r["trace"][0][1])
    R1, T1 = r["trace"][0][11:20].reshape(3, 3).astype(np.float32), r["trace"][0][20:23].astype(np.float32)
    assert int(n1) == len(model)                                  # iteration 1 pairs by index
    moved = np.empty_like(model)
    for c in range(3):                                            # transformPoints: float32, (r0 x + r1 y) + r2 z, then + T
        moved[:, c] = ((R1[c, 0] * model[:, 0] + R1[c, 1] * model[:, 1]) + R1[c, 2] * model[:, 2]) + T1[c]
    d, j = sp.cKDTree(ref.astype(np.float64)).query(moved.astype(np.float64))
    kept = int((d * d <= 3.0 * float(dm1)).sum())
    assert abs(int(r["trace"][1][0]) - kept) <= 1                 # <= 1: a pair exactly on the float32 / float64 boundary


def test_point_to_plane_model_recovers_ground_truth():
    """tests/p2plane_model.py is the yardstick of the GPU's FL_ICP_POINT_TO_PLANE tests: check the yardstick itself
    against the synthetic scene's true pose (the mode has no reference counterpart, hence no oracle)."""
    import p2plane_model as P
    from fealess_amd import synth
    Rs, ts = synth.object_pose(tx=12, ty=-8, tz=655)
    scene, _, ms = synth.render(640, 480, Rs, ts, seed=10)
    Rm = synth.rot_z(0.05) @ synth.rot_x(0.03) @ synth.rot_y(-0.04) @ Rs
    tm = ts + np.array([26.0, -16.0, 5.0])
    model, _, mm = synth.render(640, 480, Rm, tm, seed=20, noise=False, background=False)
    ys, xs = np.nonzero(ms)
    ym, xm = np.nonzero(mm)
    cw = max(xs.max() - xs.min(), xm.max() - xm.min()) + 9
    ch = max(ys.max() - ys.min(), ym.max() - ym.min()) + 9
    out = P.detection_point_to_plane(model, scene, (608.0, 608.0, 320.0, 240.0), (int(xm.min()) - 4, int(ym.min()) - 4, int(cw), int(ch)),
                                     (int(xs.min()) - 4, int(ys.min()) - 4, int(cw), int(ch)), 20, 0.0, -3.0e38,
                                     Rm.astype(np.float32), tm.astype(np.float32))
    c = (np.trace(out["R_final"].astype(np.float64) @ Rs.T) - 1) / 2
    assert np.degrees(np.arccos(np.clip(c, -1, 1))) < 1.0
    assert np.linalg.norm(out["T_final"] - ts) < 0.5
    n = out["normals"]
    has = np.linalg.norm(n, axis=1) > 0
    assert has.mean() > 0.8 and np.abs(np.linalg.norm(n[has], axis=1) - 1).max() < 1e-5
