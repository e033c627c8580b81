"""numpy/scipy model of FL_ICP_POINT_TO_PLANE (fealess_amd/csrc/fl_icp.hip), test infrastructure only.

The mode has no counterpart in the reference, so there is no oracle
for it: this file restates the kernel's arithmetic (float32 points, fp64 sums, the same gates and
loop control) so the GPU tests can check the kernel against an independent implementation, and
the tests then judge both against the ground-truth pose of a synthetic scene.
"""
import numpy as np
from scipy.spatial import cKDTree

NRM_R = 3


def scene_normals(depth, K, xs, ys):
    """Least-squares depth-gradient normals at pixels (xs, ys) of a u16 depth image (mm)."""
    fx, fy, cx, cy = K
    h, w = depth.shape
    d = depth.astype(np.float32)
    out = np.zeros((len(xs), 3), np.float32)
    r = NRM_R
    s2 = np.float32((2 * r + 1) * r * (r + 1) * (2 * r + 1) // 3)
    du, dv = np.meshgrid(np.arange(-r, r + 1, dtype=np.float32), np.arange(-r, r + 1, dtype=np.float32))
    for k, (x, y) in enumerate(zip(xs, ys)):
        if x < r or y < r or x + r >= w or y + r >= h:
            continue
        win = d[y - r:y + r + 1, x - r:x + r + 1]
        zc = d[y, x]
        gate = np.float32(0.02) * zc + np.float32(2.0)
        if zc <= 0 or (win <= 0).any() or (np.abs(win - zc) > gate).any():
            continue
        zu = np.float32((du * win).sum(dtype=np.float64)) / s2
        zv = np.float32((dv * win).sum(dtype=np.float64)) / s2
        X = np.float32((x - cx) / fx)
        Y = np.float32((y - cy) / fy)
        pu = np.array([zc / fx + X * zu, Y * zu, zu], np.float64)
        pv = np.array([X * zv, zc / fy + Y * zv, zv], np.float64)
        c = np.cross(pu, pv)
        n = np.linalg.norm(c)
        if n > 0:
            out[k] = (c / n).astype(np.float32)
    return out


def _l2dist(mod, ref, thr):
    ok = (ref[:, 2] <= 900) & (mod[:, 2] <= 900)
    d = np.sqrt(((mod.astype(np.float64) - ref.astype(np.float64)) ** 2).sum(1)).astype(np.float32)
    inl = ok & (d <= thr)
    counter = int(ok.sum())
    if counter == 0:
        return np.float32(np.finfo(np.float32).max), 0.0
    return np.float32(d[inl].astype(np.float64).sum() / max(int(inl.sum()), 1)) if inl.sum() else np.float32(np.nan), inl.sum() / counter


def rodrigues(w):
    t2 = float(w @ w)
    t = np.sqrt(t2)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    sa = np.sin(t) / t if t > 1e-9 else 1 - t2 / 6
    sb = (1 - np.cos(t)) / t2 if t > 1e-9 else 0.5 - t2 / 24
    return np.eye(3) + sa * K + sb * (K @ K)


def icp_point_to_plane(ref, nrm, model, it_thr, dmt, ddt):
    """ref/model index-paired float32 clouds (len(ref) >= len(model)); returns dict like fl_icp_result."""
    ref = np.asarray(ref, np.float32)
    nrm = np.asarray(nrm, np.float32)
    mod = np.array(model, np.float32)
    n = len(mod)
    mod[~(mod[:, 2] <= 900)] = 0
    R = np.eye(3, dtype=np.float32)
    T = np.zeros(3, np.float32)
    tree = cKDTree(ref.astype(np.float64))
    dist_mean, px = _l2dist(mod, ref[:n], np.float32(np.finfo(np.float32).max))
    dist_diff = np.float32(np.finfo(np.float32).max)
    it = 0
    n_corr = 0
    while dist_mean > dmt and dist_diff > ddt and it < it_thr:
        it += 1
        gate = np.float32(3) * dist_mean
        thr = gate * gate
        d, j = tree.query(mod.astype(np.float64))
        d2 = ((mod - ref[j]) ** 2).sum(1, dtype=np.float32)
        keep = d2 <= thr
        n_corr = int(keep.sum())
        if n_corr < 3:
            it = it_thr
            continue
        m = mod[keep].astype(np.float64)
        r = ref[j[keep]].astype(np.float64)
        nn = nrm[j[keep]].astype(np.float64)
        J = np.concatenate([np.cross(m, nn), nn], 1)
        e = (nn * (m - r)).sum(1)
        A = J.T @ J
        b = -(J.T @ e)
        tr = np.trace(A)
        A = A + 1e-12 * tr * np.eye(6)
        try:
            Lc = np.linalg.cholesky(A)
            if (np.diag(Lc) ** 2 <= 1e-13 * tr).any():
                raise np.linalg.LinAlgError
        except np.linalg.LinAlgError:
            continue
        x = np.linalg.solve(A, b)
        Ro = rodrigues(x[:3]).astype(np.float32)
        To = x[3:].astype(np.float32)
        valid = mod[:, 2] <= 900
        new = (mod @ Ro.T + To).astype(np.float32)
        mod = np.where(valid[:, None], new, mod)
        old = dist_mean
        dist_mean, px = _l2dist(mod, ref[:n], np.float32(3) * old)
        dist_diff = old - dist_mean
        T = (Ro @ T + To).astype(np.float32)
        R = (Ro @ R).astype(np.float32)
    return dict(R=R, T=T, dist_mean=float(dist_mean), px_ratio=float(px), iters=it, n_corr_last=n_corr)


def crop_pairs(model_depth, scene_depth, K, rect_model, rect_ref):
    """detection()'s paired-valid compaction (ICP/common.cpp:382-405): returns ref, mod clouds (mm) and the
    scene pixel coordinates of the kept pairs."""
    fx, fy, cx, cy = K
    mx0, my0, cw, ch = rect_model
    sx0, sy0 = rect_ref[:2]
    ys, xs = np.mgrid[0:ch, 0:cw]
    sx, sy, mx, my = sx0 + xs, sy0 + ys, mx0 + xs, my0 + ys
    zs = scene_depth[sy, sx].astype(np.float32)
    zm = model_depth[my, mx].astype(np.float32)
    A = np.stack([(sx - cx) / fx * zs, (sy - cy) / fy * zs, zs], -1).astype(np.float32)
    B = np.stack([(mx - 320.0) / 608.0 * zm, (my - 240.0) / 608.0 * zm, zm], -1).astype(np.float32)
    keep = (zs > 0) & (zm > 0) & (zs <= 900) & (zm <= 900)
    return A[keep], B[keep], sx[keep], sy[keep]


def detection_point_to_plane(model_depth, scene_depth, K, rect_model, rect_ref, it_thr, dmt, ddt, r_match, t_match):
    ref, mod, sx, sy = crop_pairs(model_depth, scene_depth, K, rect_model, rect_ref)
    nrm = scene_normals(scene_depth, K, sx, sy)
    t_tmp = (ref.astype(np.float64).mean(0) - mod.astype(np.float64).mean(0)).astype(np.float32)
    mod = (mod + t_tmp).astype(np.float32)
    t_init = t_tmp + np.asarray(t_match, np.float32)
    icp = icp_point_to_plane(ref, nrm, mod, it_thr, dmt, ddt)
    Rf = icp["R"] @ np.asarray(r_match, np.float32).reshape(3, 3)
    Tf = icp["R"] @ t_init + icp["T"]
    return dict(R_final=Rf, T_final=Tf, icp=icp, n_points=len(ref), normals=nrm)
