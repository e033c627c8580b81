"""Synthetic RGB-D frames and template banks; pure numpy.

The reference ships no data (its data/ directory is git-ignored), so every input here is
generated: an analytic scene (back plane + a sphere/box object) ray-cast to a u16 depth image in
millimetres and a shaded BGR8 image, plus template banks in three flavours:
  * random templates   (throughput: the coarse scan's work is data-independent),
  * planted templates  (copied from a frame's own quantized pyramid: guaranteed detections,
                        exercise the refinement),
  * rendered templates (views of the same object at perturbed poses, with their depth render
                        and 13-float pose, for the end-to-end Recognition path).
Feature selection for rendered templates is a simple scattered pick -- NOT the reference's
extractTemplate (training is out of scope); cropTemplates is restated in
bank.crop_templates because it defines the template geometry the matcher relies on.
"""
import numpy as np

from .bank import TemplateBank, crop_templates

FX = FY = 608.0
CX, CY = 320.0, 240.0          # initInternalMat (ICP/common.cpp:358)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def render(w, h, R_obj, t_obj, seed=0, plane_z=1200.0, noise=True, background=True, fx=FX, fy=FY, cx=CX, cy=CY):
    """Ray-cast the object (sphere radius 60 mm + box 110x70x50 mm, in object coordinates) at pose
    (R_obj, t_obj) [mm, camera frame] in front of a fronto-parallel plane.
    Returns depth (h,w) uint16 mm (0 = no return), bgr (h,w,3) uint8, mask (h,w) bool of object pixels."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)       # ray directions (z = 1)
    Rt = R_obj.T
    o = -(Rt @ t_obj)                                                            # camera origin in object frame
    dl = d @ Rt.T                                                                # directions in object frame
    best = np.full((h, w), np.inf)
    normal = np.zeros((h, w, 3))
    part = np.zeros((h, w), np.int32)
    # sphere at (-35, 0, 0), r = 60
    c_s = np.array([-35.0, 0.0, 0.0])
    r_s = 60.0
    oc = o - c_s
    a = (dl * dl).sum(-1)
    b = 2 * (dl @ oc)
    cc = oc @ oc - r_s * r_s
    disc = b * b - 4 * a * cc
    ok = disc > 0
    ts = np.where(ok, (-b - np.sqrt(np.where(ok, disc, 0))) / (2 * a), np.inf)
    ok &= ts > 0
    hit = o + dl * ts[..., None]
    n_s = (hit - c_s) / r_s
    upd = ok & (ts < best)
    best = np.where(upd, ts, best)
    normal = np.where(upd[..., None], n_s, normal)
    part = np.where(upd, 1, part)
    # box centred at (45, 0, 10), half sizes (55, 35, 25): slab method
    c_b = np.array([45.0, 0.0, 10.0])
    hs = np.array([55.0, 35.0, 25.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dl
        t1 = (c_b - hs - o) * inv
        t2 = (c_b + hs - o) * inv
    tmin = np.minimum(t1, t2)
    tmax = np.maximum(t1, t2)
    tn = tmin.max(-1)
    tf = tmax.min(-1)
    okb = (tn < tf) & (tn > 0)
    axis = tmin.argmax(-1)
    n_b = np.zeros((h, w, 3))
    sign = -np.sign(np.take_along_axis(dl, axis[..., None], -1))[..., 0]
    for k in range(3):
        n_b[..., k] = np.where(axis == k, sign, 0.0)
    upd = okb & (tn < best)
    best = np.where(upd, tn, best)
    normal = np.where(upd[..., None], n_b, normal)
    part = np.where(upd, 2, part)
    mask = np.isfinite(best)
    z_obj = best                                                                 # ray parameter = camera z (d_z = 1)
    hit_o = o + dl * np.where(mask, best, 0)[..., None]
    n_cam = normal @ R_obj.T
    depth = np.where(mask, z_obj, plane_z if background else 0.0)
    if noise:
        depth = depth + rng.integers(-1, 2, size=depth.shape) * (depth > 0)
    depth = np.clip(np.rint(depth), 0, 65535).astype(np.uint16)
    # colour: Lambert shading + object-space stripes; background gradient
    light = np.array([0.3, -0.4, -0.86])
    light /= np.linalg.norm(light)
    lam = np.clip(-(n_cam @ light), 0.05, 1.0)
    stripes = ((np.floor(hit_o[..., 0] / 14.0) + np.floor(hit_o[..., 1] / 14.0)) % 2)
    alb = np.zeros((h, w, 3))
    alb[part == 1] = (200, 120, 60)
    alb[part == 2] = (70, 160, 210)
    alb = alb * (0.55 + 0.45 * stripes[..., None])
    col = alb * lam[..., None]
    if background:
        bg = 90 + 20 * np.sin(u / 97.0) + 15 * np.cos(v / 71.0)
        col = np.where(mask[..., None], col, bg[..., None] * np.array([1.0, 0.95, 0.9]))
    if noise:
        col = col + rng.normal(0, 1.5, size=col.shape)
    bgr = np.clip(np.rint(col), 0, 255).astype(np.uint8)
    return depth, bgr, mask


def render_clutter(w, h, poses, seed=0, fx=FX, fy=FY, cx=CX, cy=CY):
    """A cluttered frame: every (R, t) of `poses` is an instance of the object (nearest surface wins per pixel), in front of
    a NON-planar background (a tilted, rippled wall 1000 - 1350 mm away with box-shaped steps) that carries a high-contrast
    texture -- colour gradients and depth normals all over the image, unlike render()'s smooth wall.
    Returns depth (h, w) uint16 mm, bgr (h, w, 3) uint8, masks (one bool image per instance, visible pixels only)."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    # background surface: tilt + two ripples + blocks of 40 - 90 mm steps
    bg_z = 1180.0 + 0.22 * (u - cx) - 0.15 * (v - cy) + 45.0 * np.sin(u / 41.0) * np.cos(v / 33.0) + 25.0 * np.sin((u + 2 * v) / 19.0)
    for _ in range(14):
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        bw, bh = int(rng.integers(30, 140)), int(rng.integers(30, 120))
        bg_z[y0:y0 + bh, x0:x0 + bw] -= float(rng.integers(40, 90))
    # background texture: checker x stripes with random per-cell albedo (edges every 12 - 20 pixels)
    cell = (np.floor(u / 17.0) + np.floor(v / 13.0)).astype(np.int64)
    alb = 60.0 + 150.0 * (((cell * 2654435761) >> 7) % 8) / 7.0
    stripes = 0.75 + 0.25 * np.sign(np.sin((u - v) / 6.0))
    bg = (alb * stripes)[..., None] * np.array([1.0, 0.9, 0.8])
    depth = bg_z.copy()
    col = bg.copy()
    masks = []
    zbuf = np.full((h, w), np.inf)
    for j, (R, t) in enumerate(poses):
        d_j, c_j, m_j = render(w, h, R, t, seed=seed * 17 + j, noise=False, background=False, fx=fx, fy=fy, cx=cx, cy=cy)
        zj = np.where(m_j, d_j.astype(np.float64), np.inf)
        win = zj < zbuf
        zbuf = np.where(win, zj, zbuf)
        depth = np.where(win, zj, depth)
        col = np.where(win[..., None], c_j.astype(np.float64), col)
        masks = [m & ~win for m in masks] + [win.copy()]
    depth = depth + rng.integers(-1, 2, size=depth.shape)
    col = col + rng.normal(0, 1.5, size=col.shape)
    return (np.clip(np.rint(depth), 0, 65535).astype(np.uint16), np.clip(np.rint(col), 0, 255).astype(np.uint8), masks)


def object_pose(tx=0.0, ty=0.0, tz=650.0, yaw=0.3, tilt=0.35, roll=0.1):
    R = rot_z(yaw) @ rot_x(tilt) @ rot_y(roll)
    return R, np.array([tx, ty, tz], np.float64)


def pose13(R, t):
    p = np.zeros(13, np.float32)
    for i in range(3):
        p[i * 4:i * 4 + 3] = R[i]
        p[i * 4 + 3] = t[i]
    p[12] = np.linalg.norm(t)
    return p


def _scatter_pick(cands, n, rng, min_dist):
    """Pick n of the candidate (x, y, label) rows, spaced apart where possible."""
    if len(cands) < n:
        return None
    order = rng.permutation(len(cands))
    chosen = []
    d = float(min_dist)
    while len(chosen) < n:
        for i in order:
            if len(chosen) >= n:
                break
            c = cands[i]
            if any((c[0] == p[0] and c[1] == p[1]) for p in chosen):
                continue
            if all((c[0] - p[0]) ** 2 + (c[1] - p[1]) ** 2 >= d * d for p in chosen):
                chosen.append(c)
        d -= 1.0
        if d < 0:
            break
    return np.array(chosen[:n], np.int32) if len(chosen) >= n else None


def features_from_quantized(q, mask, n, rng, min_dist):
    ys, xs = np.nonzero((q != 0) & mask)
    labels = np.log2(q[ys, xs].astype(np.float64)).astype(np.int32)
    cands = np.stack([xs, ys, labels], axis=1)
    return _scatter_pick(cands, n, rng, min_dist)


def quantize_pyramid(quantize_fn, bgr, depth, levels, modalities=2):
    """quantize_fn(bgr, depth, levels) -> list [l*M+m] of one-hot uint8 images (oracle or HIP)."""
    return quantize_fn(bgr, depth, levels)


def rendered_template(quantize_fn, R, t, levels, w=640, h=480, seed=0, nf0=63):
    """One template pyramid extracted from a render of the object at (R, t), plus its depth render
    in 0.1 mm and 13-float pose.  Returns (templates, pose13, depth01mm) or None."""
    rng = np.random.default_rng(seed)
    depth, bgr, mask = render(w, h, R, t, seed=seed, noise=False, background=True)
    qs = quantize_fn(bgr, depth, levels)
    M = len(qs) // levels
    tl = []
    for l in range(levels):
        m_l = mask[::2 ** l, ::2 ** l][:h >> l, :w >> l]
        n = nf0 >> l
        for m in range(M):
            f = features_from_quantized(qs[l * M + m], m_l, n, rng, min_dist=max(2.0, 10.0 / (1 << l)))
            if f is None:
                return None
            tl.append(dict(pyramid_level=l, features=f))
    tl, _ = crop_templates(tl)
    d_obj, _, _ = render(w, h, R, t, seed=seed, noise=False, background=False)
    depth01 = (d_obj.astype(np.uint32) * 10).clip(0, 65535).astype(np.uint16)
    return tl, pose13(R, t), depth01


def random_pyramid(rng, levels, M, w, h, bbox=160, nf0=63):
    """Random template pyramid (throughput banks): features uniform in a bbox x bbox window."""
    ox = int(rng.integers(0, max(1, w - bbox - 2)) // 2 * 2)
    oy = int(rng.integers(0, max(1, h - bbox - 2)) // 2 * 2)
    tl = []
    for l in range(levels):
        n = nf0 >> l
        s = bbox >> l
        for m in range(M):
            f = np.stack([rng.integers(0, s + 1, n), rng.integers(0, s + 1, n), rng.integers(0, 8, n)], axis=1)
            tl.append(dict(width=s, height=s, offset_x=ox >> l, offset_y=oy >> l, pyramid_level=l,
                           features=f.astype(np.int32)))
    return tl


def planted_pyramid(rng, qs, levels, M, w, h, bbox=160, nf0=63):
    """Template whose features are read off the frame's own quantized pyramid inside a window at a
    random position: matches that window with a (near-)perfect score."""
    for _ in range(50):
        ox = int(rng.integers(8, max(9, w - bbox - 8)) // 2 * 2)
        oy = int(rng.integers(8, max(9, h - bbox - 8)) // 2 * 2)
        tl = []
        ok = True
        for l in range(levels):
            n = nf0 >> l
            s = bbox >> l
            x0, y0 = ox >> l, oy >> l
            for m in range(M):
                q = qs[l * M + m][y0:y0 + s + 1, x0:x0 + s + 1]
                f = features_from_quantized(q, np.ones_like(q, bool), n, rng, min_dist=3.0)
                if f is None:
                    ok = False
                    break
                tl.append(dict(width=s, height=s, offset_x=x0, offset_y=y0, pyramid_level=l, features=f))
            if not ok:
                break
        if ok:
            return tl
    return None


def make_bank(class_id, n, levels, M, w, h, seed, qs=None, planted_frac=0.0, bbox=160):
    rng = np.random.default_rng(seed)
    bank = TemplateBank(class_id, levels, M)
    n_planted = int(round(n * planted_frac)) if qs is not None else 0
    planted_at = set(rng.choice(n, size=n_planted, replace=False).tolist()) if n_planted else set()
    for i in range(n):
        tl = None
        if i in planted_at:
            tl = planted_pyramid(rng, qs, levels, M, w, h, bbox)
        if tl is None:
            tl = random_pyramid(rng, levels, M, w, h, bbox)
        bank.add_pyramid(tl)
    return bank


def random_quantized(rng, w, h, density=0.35):
    """Random one-hot image (pass-through modality input)."""
    q = (1 << rng.integers(0, 8, size=(h, w))).astype(np.uint8)
    q[rng.random((h, w)) > density] = 0
    return q


def recognition_scene(quantize_fn, levels=2, w=640, h=480, seed=0, n_views=6, n_random=0, M=2):
    """A frame with the object + a bank of rendered views around the true pose (one of them close to
    it), their depth renders and poses; optionally padded with random templates.
    Returns dict(bgr, depth, bank, K, true_pose)."""
    rng = np.random.default_rng(seed)
    R_true, t_true = object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)),
                                 tz=float(rng.uniform(620, 700)), yaw=float(rng.uniform(-0.4, 0.4)),
                                 tilt=float(rng.uniform(0.25, 0.45)), roll=float(rng.uniform(-0.1, 0.2)))
    depth, bgr, _ = render(w, h, R_true, t_true, seed=seed + 1000)
    bank = TemplateBank("obj", levels, M)
    k = 0
    tries = 0
    while k < n_views and tries < 4 * n_views:
        tries += 1
        if k == 0:   # near-true view: small rotation / translation perturbation (the ICP's job)
            dR = rot_z(np.deg2rad(rng.uniform(-2, 2))) @ rot_x(np.deg2rad(rng.uniform(-2, 2)))
            R = dR @ R_true
            t = t_true + np.array([rng.uniform(-20, 20), rng.uniform(-15, 15), rng.uniform(-8, 8)])
        else:
            R, t = object_pose(tx=float(rng.uniform(-80, 80)), ty=float(rng.uniform(-50, 50)),
                               tz=float(rng.uniform(600, 760)), yaw=float(rng.uniform(-1.2, 1.2)),
                               tilt=float(rng.uniform(0.0, 0.8)), roll=float(rng.uniform(-0.5, 0.5)))
        out = rendered_template(quantize_fn, R, t, levels, w, h, seed=seed * 131 + tries)
        if out is None:
            continue
        tl, p13, d01 = out
        bank.add_pyramid(tl, p13, d01)
        k += 1
    for _ in range(n_random):              # no depth render (bank.py: pyramids without one come last)
        bank.add_pyramid(random_pyramid(rng, levels, M, w, h), None, None)
    return dict(bgr=bgr, depth=depth, bank=bank, K=(FX, FY, CX, CY), R_true=R_true, t_true=t_true)
