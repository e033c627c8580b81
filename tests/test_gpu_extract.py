"""GPU template extraction (fl_extract_template_pyramid) against the oracle's Detector::addTemplate, bit-exact
(linemod.cpp:52-164, 461-513, 747-825, 1579-1615)."""
import numpy as np
import pytest

from fealess_amd import api, synth
from fealess_amd.bank import TemplateBank

pytestmark = pytest.mark.gpu


def _render(seed, w=640, h=480, tz=650.0):
    rng = np.random.default_rng(seed)
    R, t = synth.object_pose(tx=float(rng.uniform(-50, 50)), ty=float(rng.uniform(-30, 30)), tz=tz, yaw=float(rng.uniform(-0.4, 0.4)),
                             tilt=float(rng.uniform(0.25, 0.45)), roll=float(rng.uniform(-0.1, 0.2)))
    depth, bgr, mask = synth.render(w, h, R, t, seed=seed, noise=False, background=True)
    return bgr, depth, (mask * 255).astype(np.uint8), (R, t)


def _assert_same(got, exp):
    tl_g, bb_g = got
    tl_e, feats_e, bb_e = exp
    assert tuple(bb_g) == tuple(bb_e)
    for k, t in enumerate(tl_g):
        for key in ("width", "height", "offset_x", "offset_y", "pyramid_level"):
            assert t[key] == int(tl_e[k][key]), (k, key)
        f = feats_e[k]
        assert np.array_equal(t["features"], np.stack([f["x"], f["y"], f["label"]], 1)), k


@pytest.mark.parametrize("seed,levels,use_mask", [(3, 2, True), (4, 2, False), (5, 3, True), (6, 1, True), (7, 2, True)])
def test_extract_template_pyramid_bit_exact(ctx, oracle, seed, levels, use_mask):
    bgr, depth, mask, _ = _render(seed)
    mk = mask if use_mask else None
    exp = oracle.add_template(bgr, depth, mk, levels)
    got = ctx.extract_template_pyramid(bgr, depth, mk, levels)
    assert exp is not None and got is not None
    _assert_same(got, exp)


def test_extract_fails_like_the_reference(ctx, oracle):
    flat_bgr, flat_depth = np.zeros((480, 640, 3), np.uint8), np.full((480, 640), 1000, np.uint16)
    assert oracle.add_template(flat_bgr, flat_depth, None, 2) is None
    assert ctx.extract_template_pyramid(flat_bgr, flat_depth, None, 2) is None       # addTemplate returns -1
    bgr, depth, mask, _ = _render(3)
    tiny = np.zeros_like(mask)
    tiny[200:204, 300:304] = 255                                                      # mask too small for 63 features
    assert oracle.add_template(bgr, depth, tiny, 2) is None
    assert ctx.extract_template_pyramid(bgr, depth, tiny, 2) is None


def test_extracted_templates_find_the_object(ctx, oracle):
    """Train on one render with the GPU extractor, then match a frame of the same view: the template is found where
    it was cut, with a high score -- and exactly as the oracle matcher finds it."""
    bgr, depth, mask, _ = _render(11)
    tl, bb = ctx.extract_template_pyramid(bgr, depth, mask, 2)
    bank = TemplateBank("trained", 2, 2)
    bank.add_pyramid(tl)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(bank)
    det.finalize(640, 480)
    got, n = det.match(bgr, depth, 80.0)
    exp, n_exp = oracle.match_images(bgr, depth, [5, 8], [bank], 80.0)
    assert n == n_exp and n > 0
    assert abs(int(got[0]["x"]) - bb[0]) <= 5 and abs(int(got[0]["y"]) - bb[1]) <= 5 and got[0]["similarity"] > 90.0
    for k in ("x", "y", "template_id"):
        assert np.array_equal(got[k], exp[k])
    det.close()
