"""Template bank container shared by the host API, the synthetic generator and the tests.

Mirrors the reference's data model (linemod/linemod.hpp:32-58): a class is a list of template
pyramids; a pyramid is ``levels * modalities`` Templates ordered ``[l*M + m]``
(linemod.hpp:372-373); a Template is a header plus (x, y, label) features.
"""
import numpy as np

TEMPLATE_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset_x", "<i4"), ("offset_y", "<i4"),
                           ("pyramid_level", "<i4"), ("feat_begin", "<i4"), ("feat_count", "<i4")])
FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
MATCH_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("class_idx", "<i4"),
                        ("template_id", "<i4")])


class TemplateBank:
    """One object class: n pyramids of levels*modalities templates + pose side table + depth renders."""

    def __init__(self, class_id, levels, modalities):
        self.class_id = class_id
        self.levels = levels
        self.modalities = modalities
        self._templates = []      # list of tuples
        self._features = []       # list of (n,3) int arrays
        self._nfeat = 0
        self.poses = []           # list of 13 floats
        self.model_depths = []    # model_depths[i]: (h,w) uint16 depth render (0.1 mm) of pyramid i, or None; the list may be
                                  # shorter than the bank (the pyramids behind its end have no render and cannot be refined)
        self._frozen = None

    @property
    def n_pyramids(self):
        return len(self._templates) // (self.levels * self.modalities)

    def add_pyramid(self, templates, pose13=None, model_depth=None):
        """templates: list (levels*modalities, order [l*M+m]) of dicts with width, height, offset_x,
        offset_y, pyramid_level, features (n,3) int array."""
        assert len(templates) == self.levels * self.modalities
        if model_depth is not None:
            # render i belongs to pyramid i: pyramids added without one in between get an explicit None, so that a later
            # render can never be taken for an earlier pyramid's
            self.model_depths.extend([None] * (self.n_pyramids - len(self.model_depths)))
        for t in templates:
            f = np.asarray(t["features"], dtype=np.int32).reshape(-1, 3)
            self._templates.append((t["width"], t["height"], t["offset_x"], t["offset_y"], t["pyramid_level"],
                                    self._nfeat, len(f)))
            self._features.append(f)
            self._nfeat += len(f)
        self.poses.append(np.zeros(13, np.float32) if pose13 is None else np.asarray(pose13, np.float32))
        if model_depth is not None:
            self.model_depths.append(np.ascontiguousarray(model_depth, dtype=np.uint16))
        self._frozen = None
        return self.n_pyramids - 1

    def arrays(self):
        """(templates structured array, features structured array, poses (n,13) f32)."""
        if self._frozen is None:
            t = np.array(self._templates, dtype=TEMPLATE_DTYPE) if self._templates else np.zeros(0, TEMPLATE_DTYPE)
            if self._features:
                f3 = np.concatenate(self._features, axis=0).astype(np.int32)
            else:
                f3 = np.zeros((0, 3), np.int32)
            f = np.zeros(len(f3), FEATURE_DTYPE)
            f["x"], f["y"], f["label"] = f3[:, 0], f3[:, 1], f3[:, 2]
            p = np.stack(self.poses).astype(np.float32) if self.poses else np.zeros((0, 13), np.float32)
            self._frozen = (np.ascontiguousarray(t), np.ascontiguousarray(f), np.ascontiguousarray(p))
        return self._frozen

    def subset(self, first, count):
        """Contiguous shard [first, first+count) as a new bank (template ids restart at 0)."""
        t, f, p = self.arrays()
        LM = self.levels * self.modalities
        out = TemplateBank(self.class_id, self.levels, self.modalities)
        for i in range(first, first + count):
            tl = []
            for k in range(LM):
                h = t[i * LM + k]
                fr = f[h["feat_begin"]:h["feat_begin"] + h["feat_count"]]
                tl.append(dict(width=int(h["width"]), height=int(h["height"]), offset_x=int(h["offset_x"]),
                               offset_y=int(h["offset_y"]), pyramid_level=int(h["pyramid_level"]),
                               features=np.stack([fr["x"], fr["y"], fr["label"]], axis=1)))
            md = self.model_depths[i] if i < len(self.model_depths) else None
            out.add_pyramid(tl, p[i], md)
        return out


def crop_templates(templates):
    """cropTemplates (linemod/linemod.cpp:52-96) restated for the synthetic generators: shift the
    features of all levels/modalities of one view to a common bounding box and fill in
    width/height/offset.  `templates` is a list of dicts with pyramid_level and absolute features."""
    min_x = min_y = np.iinfo(np.int32).max
    max_x = max_y = np.iinfo(np.int32).min
    for t in templates:
        f = np.asarray(t["features"]).reshape(-1, 3)
        if len(f) == 0:
            continue
        lv = t["pyramid_level"]
        min_x = min(min_x, int((f[:, 0] << lv).min()))
        min_y = min(min_y, int((f[:, 1] << lv).min()))
        max_x = max(max_x, int((f[:, 0] << lv).max()))
        max_y = max(max_y, int((f[:, 1] << lv).max()))
    if min_x % 2 == 1:
        min_x -= 1
    if min_y % 2 == 1:
        min_y -= 1
    out = []
    for t in templates:
        lv = t["pyramid_level"]
        f = np.asarray(t["features"], dtype=np.int64).reshape(-1, 3).copy()
        ox, oy = min_x >> lv, min_y >> lv
        f[:, 0] -= ox
        f[:, 1] -= oy
        out.append(dict(width=(max_x - min_x) >> lv, height=(max_y - min_y) >> lv, offset_x=ox, offset_y=oy,
                        pyramid_level=lv, features=f.astype(np.int32)))
    return out, (min_x, min_y, max_x - min_x, max_y - min_y)
