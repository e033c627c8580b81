#!/usr/bin/env python3
"""Feasibility model (CPU, numpy + cKDTree) for the ICP search cache of fl_icp.hip: how many exact-NN searches per
iteration can be skipped when each model point keeps (partner j, lower bound L on the distance to every OTHER
reference point)?  L decays by the point's motion every iteration; a search is needed only when the recomputed
distance to j is not strictly below L.  Prints the per-iteration fraction of queries that still need a search, for
bench-like scenes (fealess_amd.synth, template view perturbed like bench.py's)."""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fealess_amd import synth  # noqa: E402


def backproject(depth, rect, fx=608.0, fy=608.0, cx=320.0, cy=240.0):
    x0, y0, w, h = rect
    u, v = np.meshgrid(np.arange(x0, x0 + w, dtype=np.float64), np.arange(y0, y0 + h, dtype=np.float64))
    z = depth[y0:y0 + h, x0:x0 + w].astype(np.float64)
    z = np.where(z == 0, np.nan, z)
    return np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1).reshape(-1, 3)


def clouds(seed):
    rng = np.random.default_rng(seed)
    R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)), tz=float(rng.uniform(620, 700)),
                             yaw=float(rng.uniform(-0.4, 0.4)), tilt=float(rng.uniform(0.25, 0.45)),
                             roll=float(rng.uniform(-0.1, 0.2)))
    depth, _, mask = synth.render(640, 480, R, t, seed=100 + seed)
    dR = synth.rot_z(np.deg2rad(rng.uniform(-2, 2))) @ synth.rot_x(np.deg2rad(rng.uniform(-2, 2)))
    tt = t + np.array([rng.uniform(-20, 20), rng.uniform(-15, 15), rng.uniform(-8, 8)])
    dm, _, mm = synth.render(640, 480, dR @ R, tt, seed=1000 + seed, noise=False, background=False)
    ys, xs = np.nonzero(mm)
    rm = (xs.min(), ys.min(), xs.max() - xs.min(), ys.max() - ys.min())
    ys, xs = np.nonzero(mask)
    # the match lands on a multiple of T = 5 near the object's corner
    rr = (int(xs.min() // 5 * 5), int(ys.min() // 5 * 5), rm[2], rm[3])
    ref = backproject(depth, rr)
    mod = backproject(dm, rm)
    ok = (ref[:, 2] <= 900) & (mod[:, 2] <= 900)
    ref, mod = ref[ok], mod[ok]
    mod = mod + (ref.mean(0) - mod.mean(0))
    return ref, mod


def run(seed, iters=20, grow=1.0):
    ref, mod = clouds(seed)
    n = len(mod)
    tree = cKDTree(ref)
    dist_mean = np.linalg.norm(mod - ref, axis=1).mean()
    j = np.full(n, -1)
    L = np.zeros(n)                     # lower bound on the distance to every reference point other than j
    rows = []
    for it in range(1, iters + 1):
        thr = 3 * dist_mean
        if it == 1:
            pm, pr = mod, ref           # index pairs
            searched = 0
        else:
            dj = np.where(j >= 0, np.linalg.norm(mod - ref[np.maximum(j, 0)], axis=1), np.inf)
            r_thr = np.sqrt(thr)
            sure_nn = dj * 1.00001 < L                              # partner is still the strict nearest neighbour
            sure_drop = (np.minimum(dj, L) > r_thr * 1.00001)       # whatever the NN is, it is beyond the gate
            need = ~(sure_nn | sure_drop)
            searched = int(need.sum())
            if searched:
                q = mod[need]
                # window radius the kernel would use (bound on the NN distance, gate), optionally grown to buy a better L
                r = np.minimum(dj[need], r_thr) * grow + 0.5
                d2, i2 = tree.query(q, k=2)
                j[need] = i2[:, 0]
                # L: second nearest seen, but nothing outside the window is known -> min(d2, r)
                L[need] = np.minimum(d2[:, 1], r)
            dj = np.linalg.norm(mod - ref[np.maximum(j, 0)], axis=1)
            keep = dj * dj <= thr
            pm, pr = mod[keep], ref[j[keep]]
        mc, rc = pm.mean(0), pr.mean(0)
        C = pm.T @ pr
        U, _, Vt = np.linalg.svd(C)
        Ro = Vt.T @ U.T
        To = rc - Ro @ mc
        new = mod @ Ro.T + To
        move = np.linalg.norm(new - mod, axis=1)
        mod = new
        L = L - move * 1.00001
        d = np.linalg.norm(mod - ref, axis=1)
        inl = d <= 3 * dist_mean
        dist_mean = d[inl].mean()
        rows.append((it, searched / n, move.max(), dist_mean))
    return n, rows


if __name__ == "__main__":
    grow = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    tot = np.zeros(21)
    for seed in range(6):
        n, rows = run(seed, grow=grow)
        print(f"scene {seed}: n = {n}")
        for it, frac, mv, dm in rows:
            tot[it] += frac
            print(f"  iter {it:2d}: searched {frac * 100:5.1f} %   max move {mv:8.4f} mm   dist_mean {dm:7.4f}")
    print("mean searched fraction over iterations 2..20: %.3f" % (tot[2:].sum() / 6 / 19))
