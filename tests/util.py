"""Shared helpers for the test-suite (fixture loading, naive Python re-derivations)."""
import os

import numpy as np

from fealess_amd.bank import TemplateBank

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bank_from_arrays(templates, features, poses, levels, modalities, class_id="obj", model_depths=None):
    b = TemplateBank(class_id, levels, modalities)
    LM = levels * modalities
    n = len(templates) // LM
    for i in range(n):
        tl = []
        for k in range(LM):
            h = templates[i * LM + k]
            fr = features[int(h["feat_begin"]):int(h["feat_begin"]) + int(h["feat_count"])]
            tl.append(dict(width=int(h["width"]), height=int(h["height"]), offset_x=int(h["offset_x"]),
                           offset_y=int(h["offset_y"]), pyramid_level=int(h["pyramid_level"]),
                           features=np.stack([fr["x"], fr["y"], fr["label"]], axis=1).astype(np.int32)))
        b.add_pyramid(tl, poses[i] if len(poses) else None, None if model_depths is None else model_depths[i])
    return b


# ---- naive restatements used as an INDEPENDENT check of the oracle on tiny inputs ----------------
def naive_spread(q, T):
    h, w = q.shape
    out = np.zeros_like(q)
    for y in range(h):
        for x in range(w):
            v = 0
            for r in range(T):
                for c in range(T):
                    if y + r < h and x + c < w:
                        v |= int(q[y + r, x + c])
            out[y, x] = v
    return out


def naive_response(spread_byte, ori):
    """max over set bits of g(circular distance) with g = 4, 2, 1, 0 (this fork's LUT, linemod.cpp:970)."""
    best = 0
    for b in range(8):
        if spread_byte & (1 << b):
            d = min((b - ori) % 8, (ori - b) % 8)
            best = max(best, {0: 4, 1: 2, 2: 1}.get(d, 0))
    return best


def naive_similarity(q, T, feats, width, height):
    """similarity() (linemod.cpp:1130-1214) written WITHOUT linear memories: response maps are
    addressed in image coordinates with the reference's 1-D wrap-around made explicit."""
    h, w = q.shape
    W, H = w // T, h // T
    sp = naive_spread(q, T)
    wf, hf = (width - 1) // T + 1, (height - 1) // T + 1
    P = (H - hf) * W + (W - wf) + 1
    dst = np.zeros(W * H, np.int64)
    for (fx, fy, lab) in feats:
        if fx < 0 or fx >= w or fy < 0 or fy >= h:
            continue
        gx, gy = fx % T, fy % T
        base = (fy // T) * W + fx // T
        for j in range(max(P, 0)):
            k = base + j
            # position k of the linear memory of grid (gy, gx); past the end it continues into
            # the next grid cell's memory (continuous Mat), past the last one it is zero here
            g = gy * T + gx
            while k >= W * H and g < T * T - 1:
                k -= W * H
                g += 1
            if k >= W * H:
                continue
            ggy, ggx = g // T, g % T
            yy, xx = (k // W) * T + ggy, (k % W) * T + ggx
            dst[j] += naive_response(int(sp[yy, xx]), lab)
    return dst.reshape(H, W)
