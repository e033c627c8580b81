"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/fealess_hip.h declares, fails loudly without a GPU (no CPU fallback), and the host-only
entry points (fl_lm_label_stride, fl_merge_topk) behave.  Also the CadReco adapter's bank I/O
(OpenCV FileStorage YAML subset + 16-bit PNG), which needs no GPU."""
import ctypes as C
import os
import re
import struct
import zlib

import numpy as np
import pytest

from fealess_amd import _lib as L
from fealess_amd.bank import MATCH_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_header_symbols_all_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "fealess_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in fealess_hip.h but not exported"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    assert lib.fl_abi_version() == 1


def test_mg_header_symbols_all_exported_and_bound():
    """include/fealess_mg.h (the C++ multi-GPU host on RCCL): the library loads without a GPU and exports what it declares."""
    hdr = open(os.path.join(ROOT, "include", "fealess_mg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fl_mg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 5
    lib = L.load_mg()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in fealess_mg.h but not exported"
    assert declared == set(L.MG_SIGNATURES), declared ^ set(L.MG_SIGNATURES)
    assert C.sizeof(L.MgResult) == 4 + 4 + 20 + 64
    assert lib.fl_mg_create(None, None, 1, 0, 0, 0, 1, C.byref(C.c_void_p())) == L.FL_ERR_INVALID     # argument checks need no GPU


def test_no_cpu_fallback():
    if _has_gpu():
        pytest.skip("a GPU is present")
    lib = L.load()
    h = C.c_void_p()
    assert lib.fl_context_create(0, C.byref(h)) == L.FL_ERR_NO_DEVICE and not h.value
    from fealess_amd import api
    with pytest.raises(api.FealessError):
        api.Context(0)


def test_product_does_not_touch_the_oracle():
    """The product path must never import, link or load anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fealess_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "oracle_py" not in txt and "fealess_oracle.h" not in txt, (dirpath, f)


def test_search_bound_packing_saturates():
    """The ICP kernel keeps its per-point search bound as the top 16 bits of a float32, rounded UP (bnd_st / bnd_ld in
    fl_icp.hip; the packing is host + device code, exported as fl_dev_bnd_pack for this check).  A stored bound may be larger
    than the value, never smaller; +inf stays +inf, FLT_MAX rounds up to +inf, and EVERY NaN payload -- also the ones whose low
    mantissa bits would carry into the exponent or the sign -- is stored as a NaN ("no bound"), never as a small number."""
    lib = L.load()
    lib.fl_dev_bnd_pack.restype = C.c_uint
    lib.fl_dev_bnd_pack.argtypes = [C.c_uint]

    def unpack(p):
        return np.array([p << 16], np.uint32).view(np.float32)[0]

    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.uniform(0, 1e4, 2000), 10.0 ** rng.uniform(-30, 38, 2000), [0.0, 1.0, 3.4028235e38, 65536.0, 1e-45]]).astype(np.float32)
    for v in vals:
        got = unpack(lib.fl_dev_bnd_pack(int(np.array([v], np.float32).view(np.uint32)[0])))
        assert got >= v and (np.isinf(got) or got <= v * np.float32(1.008) + np.float32(1e-38)), (v, got)
    assert np.isinf(unpack(lib.fl_dev_bnd_pack(0x7F800000))) and np.isinf(unpack(lib.fl_dev_bnd_pack(0x7F7FFFFF)))
    for bits in (0x7FC00000, 0x7FFFFFFF, 0x7FFF0001, 0x7F800001, 0x7F80FFFF, 0xFFC00000, 0xFFFFFFFF, 0xFFFF0001, 0xFF800001, 0x80000000, 0xBF800000):
        assert np.isnan(unpack(lib.fl_dev_bnd_pack(bits))), hex(bits)     # NaNs and (impossible) negative bounds: no bound


def test_lm_label_stride_matches_oracle(oracle):
    lib = L.load()
    for (w, h, T) in [(640, 480, 5), (320, 240, 8), (1280, 720, 5), (320, 180, 4), (64, 48, 8)]:
        assert lib.fl_lm_label_stride(w, h, T) == oracle.lib().orc_lm_label_stride(w, h, T)


def test_merge_topk_equals_global_sort_unique(oracle):
    rng = np.random.default_rng(0)
    n = 400
    rec = np.zeros(n, MATCH_DTYPE)
    rec["x"] = rng.integers(0, 8, n) * 5
    rec["y"] = rng.integers(0, 6, n) * 5
    rec["similarity"] = rng.integers(160, 200, n) / np.float32(2.0)
    rec["template_id"] = rng.integers(0, 50, n)
    rec["template_id"][::17] = -1            # padding records of short shards
    from fealess_amd.api import merge_topk
    got = merge_topk(rec, n)
    live = rec[rec["template_id"] >= 0].copy()
    live = np.ascontiguousarray(live)
    k = oracle.lib().orc_sort_unique(live.ctypes.data_as(C.c_void_p), len(live))
    assert len(got) == k and got.tobytes() == live[:k].tobytes()


# ---- CadReco adapter bank I/O --------------------------------------------------------------------
def _cad():
    path = os.path.join(ROOT, "fealess_amd", "cadreco", "libcadreco_hip.so")
    lib = C.CDLL(path)
    lib.cadreco_version.restype = C.c_char_p
    return lib


def write_png16(path, img):
    img = np.ascontiguousarray(img, dtype=">u2")
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_linemod_yaml(path, bank, T, modalities=("ColorGradient", "DepthNormal")):
    """OpenCV FileStorage YAML 1.0 as cup_linemod's writeLinemod lays it out (linemod_if.cpp:49-63)."""
    t, f, p = bank.arrays()
    LM = bank.levels * bank.modalities
    o = ["%YAML:1.0", "---", f"pyramid_levels: {bank.levels}", "T: [ " + ", ".join(str(v) for v in T) + " ]", "modalities:"]
    for m in modalities[:bank.modalities]:
        if m == "ColorGradient":
            o += ["   -", "      type: ColorGradient", "      weak_threshold: 10.", "      num_features: 63", "      strong_threshold: 55."]
        else:
            o += ["   -", "      type: DepthNormal", "      distance_threshold: 2000", "      difference_threshold: 50",
                  "      num_features: 63", "      extract_threshold: 2"]
    o += ["classes:", "   -", f'      class_id: "{bank.class_id}"',
          "      modalities: [ " + ", ".join(modalities[:bank.modalities]) + " ]", f"      pyramid_levels: {bank.levels}",
          "      template_pyramids:"]
    for i in range(bank.n_pyramids):
        pose = [repr(float(v)).rstrip("0") if "." in repr(float(v)) else repr(float(v)) for v in p[i]]
        o += ["         -", f"            template_id: {i}",
              "            template_pose: [ " + ", ".join(pose[:7]) + ",", "               " + ", ".join(pose[7:]) + " ]",
              "            templates:"]
        for k in range(LM):
            hdr = t[i * LM + k]
            o += ["               -", f"                  width: {hdr['width']}", f"                  height: {hdr['height']}",
                  f"                  offset_x: {hdr['offset_x']}", f"                  offset_y: {hdr['offset_y']}",
                  f"                  pyramid_level: {hdr['pyramid_level']}", "                  features:"]
            for fr in f[hdr["feat_begin"]:hdr["feat_begin"] + hdr["feat_count"]]:
                o.append(f"                     - [ {fr['x']}, {fr['y']}, {fr['label']} ]")
    with open(path, "w") as fh:
        fh.write("\n".join(o) + "\n")


def test_png16_reader(tmp_path):
    lib = _cad()
    img = np.random.default_rng(1).integers(0, 65536, (37, 53)).astype(np.uint16)
    p = str(tmp_path / "d.png")
    write_png16(p, img)
    out = np.zeros(img.size, np.uint16)
    w, h = C.c_int(0), C.c_int(0)
    assert lib.cadreco_read_png16(p.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h)) == 0
    assert (w.value, h.value) == (53, 37) and np.array_equal(out.reshape(37, 53), img)
    assert lib.cadreco_read_png16(str(tmp_path / "missing.png").encode(), out.ctypes.data_as(C.c_void_p), out.size,
                                  C.byref(w), C.byref(h)) == -1


def test_png16_reader_rejects_malformed_files(tmp_path):
    """The depth renders are file input: a short IHDR at the end of the file, a truncated chunk, a truncated IDAT stream and
    an absurd size must be refused, not read out of bounds."""
    lib = _cad()
    img = np.arange(20 * 30, dtype=np.uint16).reshape(20, 30)
    good = str(tmp_path / "g.png")
    write_png16(good, img)
    data = open(good, "rb").read()
    out = np.zeros(img.size, np.uint16)
    w, h = C.c_int(0), C.c_int(0)

    def read(blob):
        p = str(tmp_path / "x.png")
        open(p, "wb").write(blob)
        return lib.cadreco_read_png16(p.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(w), C.byref(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    assert read(data) == 0
    assert read(data[:8] + chunk(b"IHDR", b"\x00\x00\x00\x1e\x00\x00")) == -1          # IHDR of 6 bytes as the last chunk
    assert read(data[:40]) == -1                                                        # chunk cut short
    assert read(data[:-30] + data[-12:]) == -1                                          # IDAT stream does not inflate to w*h
    huge = chunk(b"IHDR", struct.pack(">IIBBBBB", 1 << 20, 1 << 20, 16, 0, 0, 0, 0))
    assert read(data[:8] + huge + data[8 + 25:]) == -1                                  # 2^40 pixels


def test_addobj_rejects_inconsistent_bank_files(tmp_path):
    """AddObj hands T[0..levels) and one pose per pyramid to the C ABI: a file whose T is shorter than pyramid_levels must be
    refused before that (no GPU needed: the check precedes any device work; without a device AddObj fails anyway)."""
    from fealess_amd import synth
    lib = _cad()
    lib.cadreco_create.restype = C.c_void_p
    bank = synth.make_bank("obj", 3, 2, 2, 640, 480, seed=3)
    d = tmp_path / "bank"
    d.mkdir()
    write_linemod_yaml(str(d / "linemod_templates.yml"), bank, [5])          # T has 1 entry, pyramid_levels says 2
    h = lib.cadreco_create(1)
    assert h
    rc = lib.cadreco_add_obj(C.c_void_p(h), str(d).encode())
    assert rc != 0
    lib.cadreco_destroy(C.c_void_p(h))


def test_linemod_yaml_reader(tmp_path):
    from fealess_amd import synth
    lib = _cad()
    bank = synth.make_bank("c919-jig", 7, 2, 2, 640, 480, seed=3)
    p = str(tmp_path / "linemod_templates.yml")
    write_linemod_yaml(p, bank, [5, 8])
    lv, nc, nt, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert lib.cadreco_read_linemod(p.encode(), C.byref(lv), C.byref(nc), C.byref(nt), C.byref(nf)) == 0
    assert (lv.value, nc.value, nt.value, nf.value) == (2, 1, 7, 7 * 2 * (63 + 31))
    assert b"HIP" in lib.cadreco_version()
    # packed binary cache next to the YAML: written on the first read, used on the second,
    # ignored again as soon as the YAML changes
    fc = C.c_int(-1)
    assert lib.cadreco_read_linemod_cached(p.encode(), C.byref(fc), C.byref(nt), C.byref(nf)) == 0
    assert fc.value == 0 and os.path.exists(p + ".flbank") and (nt.value, nf.value) == (7, 7 * 2 * (63 + 31))
    assert lib.cadreco_read_linemod_cached(p.encode(), C.byref(fc), C.byref(nt), C.byref(nf)) == 0
    assert fc.value == 1 and (nt.value, nf.value) == (7, 7 * 2 * (63 + 31))
    bank2 = synth.make_bank("c919-jig", 9, 2, 2, 640, 480, seed=4)
    write_linemod_yaml(p, bank2, [5, 8])
    os.utime(p, (1, 1))                                      # a different mtime even within the same second
    assert lib.cadreco_read_linemod_cached(p.encode(), C.byref(fc), C.byref(nt), C.byref(nf)) == 0
    assert fc.value == 0 and nt.value == 9
    with open(p + ".flbank", "r+b") as f:                    # a truncated / corrupt cache is ignored, not trusted
        f.truncate(40)
    assert lib.cadreco_read_linemod_cached(p.encode(), C.byref(fc), C.byref(nt), C.byref(nf)) == 0
    assert fc.value == 0 and nt.value == 9


def test_cadreco_factory_rejects_unsupported_types():
    lib = _cad()
    lib.cadreco_create.restype = C.c_void_p
    for unsupported in (0, 2, 3):            # EObjReco_FEATURE, BB8, PoseNet -> nullptr (obj_reco_temp.cpp:13-30)
        assert not lib.cadreco_create(unsupported)


def test_bank_depth_renders_stay_with_their_pyramids():
    """model_depths[i] is the render of pyramid i: a render added after pyramids without one must not slide down to an
    earlier pyramid (api.finalize uploads by index, subset() copies by index)."""
    from fealess_amd import synth
    from fealess_amd.bank import TemplateBank
    rng = np.random.default_rng(0)
    bank = TemplateBank("obj", 2, 2)
    d0 = np.full((480, 640), 7, np.uint16)
    d3 = np.full((480, 640), 9, np.uint16)
    bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, d0)
    bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, None)
    bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, None)
    bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, d3)
    bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, None)
    assert bank.n_pyramids == 5 and len(bank.model_depths) == 4
    assert bank.model_depths[0][0, 0] == 7 and bank.model_depths[1] is None and bank.model_depths[2] is None and bank.model_depths[3][0, 0] == 9
    sub = bank.subset(2, 3)
    assert sub.n_pyramids == 3 and sub.model_depths[0] is None and sub.model_depths[1][0, 0] == 9 and len(sub.model_depths) == 2


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no WORLD_SIZE) must run TWO ranks: the parent starts torch.distributed.run
    as a child and relays rank 0's line; inside a rank a WORLD_SIZE that differs from --gpus fails loudly.  --launch-check
    keeps the GPU out of it (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collectives"]["ranks"] == 2
    env["WORLD_SIZE"] = "1"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr
