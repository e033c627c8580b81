"""GPU: the HIP path vs the committed golden fixtures (data only -- nothing here needs
/root/reference), the C++ CadReco facade end to end on a data directory written in the
reference's on-disk format, and the template-sharded top-k export/merge on one GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from fealess_amd import api, synth
from fealess_amd import _lib as L
from fealess_amd.bank import MATCH_DTYPE
from test_abi_cpu import write_linemod_yaml, write_png16

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_linemod(ctx):
    g = util.golden("linemod_320x160.npz")
    bank = util.bank_from_arrays(g["templates"], g["features"], g["poses"], 2, 2)
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(bank)
    det.finalize(320, 160)
    m, n = det.match_quantized([g["q0"], g["q1"], g["q2"], g["q3"]], float(g["threshold"]))
    assert n == int(g["n_matches"]) and m.tobytes() == g["matches"].tobytes()
    assert np.array_equal(det.similarity_maps(0, bank.n_pyramids), g["sims"])
    lm0 = ctx.build_linear_memories(g["q0"], 5)
    crc = [int(lm0.astype(np.uint64).sum()),
           int((lm0.astype(np.uint64) * (np.arange(lm0.size, dtype=np.uint64).reshape(lm0.shape) % 251)).sum())]
    assert crc == g["lm_level0_mod0_crc"].tolist()
    det.close()


def test_golden_frontend(ctx):
    g = util.golden("frontend_256x192.npz")
    assert np.array_equal(ctx.quantized_orientations(g["bgr"], 10.0), g["qo"])
    assert np.array_equal(ctx.quantized_normals(g["depth"]), g["qn"])
    assert np.array_equal(ctx.pyrdown_bgr(g["bgr"]), g["pyrdown"])
    assert np.array_equal(ctx.quantized_orientations(g["pyrdown"], 10.0), g["qo1"])


def test_golden_icp(ctx):
    g = util.golden("icp_1500.npz")
    r = ctx.icp_cloud_to_cloud_ex(g["ref"], g["model"], 12, 0.0, -3.0e38, L.FL_ICP_PARITY)
    assert np.array_equal(r["R"], g["R32"]) and np.array_equal(r["T"], g["T32"]) and r["dist_mean"] == g["dm32"]
    r = ctx.icp_cloud_to_cloud_ex(g["ref"], g["model"], 12, 0.0, -3.0e38, L.FL_ICP_FAST)
    assert np.abs(r["R"] - g["R64"]).max() <= 1e-4 and np.abs(r["T"] - g["T64"]).max() <= 1e-3
    r = ctx.icp_cloud_to_cloud_ex(g["ref"], g["model"], 10, 0.5, 0.01, L.FL_ICP_PARITY)
    assert r["iters"] == int(g["iters_def"]) and np.array_equal(r["R"], g["Rdef"]) and np.array_equal(r["T"], g["Tdef"])


def test_golden_recognition(ctx):
    g = util.golden("recognition_vga.npz")
    bank = util.bank_from_arrays(g["templates"], g["features"], g["poses"], 2, 2, model_depths=g["model_depths"])
    det = api.Detector(ctx, 2, [5, 8])
    det.add_class(bank)
    det.finalize(640, 480)
    K = tuple(float(v) for v in g["K"])
    r = det.recognize_batch([g["bgr"]], [g["depth"]], K, 75.0, 10, 0.5, 0.01)[0]
    assert r["found"] == 1 and r["n_matches"] == int(g["n_matches"])
    assert [r["best"]["x"], r["best"]["y"], r["best"]["template_id"]] == g["best"].tolist()
    assert r["best"]["similarity"] == g["best_sim"] and r["det"]["n_points"] == int(g["n_points"])
    assert np.abs(r["pose"] - g["pose"]).max() <= 1e-4 and np.array_equal(r["pose"], g["pose"])
    r20 = det.recognize_batch([g["bgr"]], [g["depth"]], K, 75.0, 20, -1.0, -3.0e38)[0]
    assert r20["det"]["icp"]["iters"] == 20 and np.array_equal(r20["pose"], g["pose20"])
    det.close()


def test_cadreco_facade_end_to_end(tmp_path, oracle):
    """CObjRecoCAD::Create -> AddObj(dir) -> Recognition on a directory in the reference's format
    (linemod_templates.yml + depth/<id>.png, obj_reco_lmicp.cpp:67-74,156-157)."""
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=13, n_views=3)
    d = tmp_path / "obj"
    (d / "depth").mkdir(parents=True)
    write_linemod_yaml(str(d / "linemod_templates.yml"), sc["bank"], [5, 8])
    for i, md in enumerate(sc["bank"].model_depths):
        write_png16(str(d / "depth" / f"{i}.png"), md)
    lib = C.CDLL(os.path.join(ROOT, "fealess_amd", "cadreco", "libcadreco_hip.so"))
    lib.cadreco_create.restype = C.c_void_p
    h = C.c_void_p(lib.cadreco_create(1))                   # EObjReco_LmICP
    assert h.value
    assert lib.cadreco_add_obj(h, str(tmp_path / "nope").encode()) == C.c_int(0x80000002).value   # ERROR_OPEN_FILE_FAILED
    assert lib.cadreco_add_obj(h, str(d).encode()) == 0
    bgr, depth = np.ascontiguousarray(sc["bgr"]), np.ascontiguousarray(sc["depth"])
    pose = np.zeros(16, np.float32)
    tag = C.create_string_buffer(64)
    n = C.c_int(-1)
    fx, fy, cx, cy = sc["K"]

    def reco(kw=640, kh=480, ts=1.0):
        return lib.cadreco_recognition(h, bgr.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), 640, 480,
                                       C.c_double(ts), kw, kh, C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy),
                                       C.byref(n), pose.ctypes.data_as(C.c_void_p), tag, 64)
    assert reco() == 0 and n.value == 1 and tag.value == b"obj"
    exp = oracle.recognition(bgr, depth, sc["K"], [5, 8], sc["bank"], 75.0, 10, 0.5, 0.01)   # constructor defaults :52-55
    assert np.abs(pose.reshape(4, 4) - exp["pose"]).max() <= 1e-4
    assert pose.reshape(4, 4)[3].tolist() == [0, 0, 0, 1]
    assert reco(kw=320) == C.c_int(0x80000001).value         # size != intrinsics size -> ERROR_INVALID_PARAM (:223-227)
    assert reco(ts=-1.0) == C.c_int(0x80000001).value        # negative timestamp (CheckTImage :35)
    # a 1280x960 frame is zoomed to 640x480 with cv::resize(INTER_LINEAR) (:229-249) while detection() still gets
    # the caller's un-zoomed intrinsics (:190).  Pixel replication makes the zoomed frame equal the 640x480 one.
    bgr2 = np.ascontiguousarray(np.repeat(np.repeat(bgr, 2, axis=0), 2, axis=1))
    depth2 = np.ascontiguousarray(np.repeat(np.repeat(depth, 2, axis=0), 2, axis=1))
    rc = lib.cadreco_recognition(h, bgr2.ctypes.data_as(C.c_void_p), depth2.ctypes.data_as(C.c_void_p), 1280, 960, C.c_double(1.0),
                                 1280, 960, C.c_double(2 * fx), C.c_double(2 * fy), C.c_double(2 * cx), C.c_double(2 * cy),
                                 C.byref(n), pose.ctypes.data_as(C.c_void_p), tag, 64)
    assert rc == 0 and n.value == 1
    exp2 = oracle.recognition(bgr, depth, (2 * fx, 2 * fy, 2 * cx, 2 * cy), [5, 8], sc["bank"], 75.0, 10, 0.5, 0.01)
    assert np.abs(pose.reshape(4, 4) - exp2["pose"]).max() <= 1e-4
    assert reco() == 0 and np.abs(pose.reshape(4, 4) - exp["pose"]).max() <= 1e-4      # back to 640x480 (re-finalize not needed)
    # opt-in multi-hypothesis mode (CadRecoSetMultiHypothesis): the first k matches refined, NMS winners returned, best first
    lib.cadreco_set_multi_hypothesis.argtypes = [C.c_void_p, C.c_int, C.c_float]
    assert lib.cadreco_set_multi_hypothesis(h, 0, 20.0) == C.c_int(0x80000001).value
    assert lib.cadreco_set_multi_hypothesis(h, 6, 20.0) == 0
    lib.cadreco_set_params.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int]
    assert lib.cadreco_set_params(h, 60.0, 8, 0.3, 0.01, 0) == 0
    poses = np.zeros((8, 16), np.float32)
    rc = lib.cadreco_recognition_all(h, bgr.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), 640, 480, C.c_double(1.0),
                                     640, 480, C.c_double(fx), C.c_double(fy), C.c_double(cx), C.c_double(cy), C.byref(n),
                                     poses.ctypes.data_as(C.c_void_p), 8)
    hyp, win = oracle.recognition_topk(bgr, depth, sc["K"], [5, 8], sc["bank"], 6, 60.0, 8, 0.3, 0.01, nms_dist=20.0)
    want = [hyp[i]["pose"] for i in win if hyp[i]["found"]]
    assert rc == 0 and n.value == len(want) >= 1
    for i, wp in enumerate(want):
        assert np.abs(poses[i].reshape(4, 4) - wp).max() <= 1e-4
    assert lib.cadreco_set_multi_hypothesis(h, 1, 20.0) == 0              # back to the reference behaviour: matches[0] only
    assert lib.cadreco_set_params(h, 75.0, 10, 0.5, 0.01, 0) == 0
    assert reco() == 0 and n.value == 1 and np.abs(pose.reshape(4, 4) - exp["pose"]).max() <= 1e-4
    lib.cadreco_destroy(h)
    # the same through a C++ caller that holds a CObjRecoCAD* (virtual calls, std::string / std::vector across the boundary)
    import subprocess
    cad = os.path.join(ROOT, "fealess_amd", "cadreco")
    exe = str(tmp_path / "caller")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-I", cad, os.path.join(ROOT, "tests", "dropin", "tu_caller_gpu.cpp"), "-o", exe,
                        "-L", cad, "-lcadreco_hip", "-Wl,-rpath," + cad, "-Wl,-rpath," + os.path.join(ROOT, "fealess_amd", "csrc")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    bgr.tofile(str(tmp_path / "f.bgr"))
    depth.tofile(str(tmp_path / "f.d16"))
    r = subprocess.run([exe, str(d), str(tmp_path / "f.bgr"), str(tmp_path / "f.d16"), "640", "480", repr(fx), repr(fy), repr(cx), repr(cy)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    lines = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
    assert lines["addobj"] == "0" and lines["recognition"] == "0 1" and lines["tag"] == "obj"
    cpose = np.array([float(v) for v in lines["pose"].split()], np.float32).reshape(4, 4)
    assert np.array_equal(cpose, np.asarray(exp["pose"], np.float32)) or np.abs(cpose - exp["pose"]).max() <= 1e-4


def test_template_sharded_topk_on_one_gpu(ctx, oracle):
    """Two 'ranks' = two detectors holding the two halves of a bank; export_topk + merge must equal
    the single-detector match over the whole bank (C4 of BASELINE.json, minus the wire)."""
    import torch
    from fealess_amd import distributed as D
    rng = np.random.default_rng(31)
    w0, h0, T = 640, 480, [5, 8]
    qs = [synth.random_quantized(rng, w0 >> l, h0 >> l, 0.03) for l in range(2) for _ in range(2)]
    bank = synth.make_bank("obj", 60, 2, 2, w0, h0, seed=4, qs=qs, planted_frac=0.3)
    k = 32
    bufs = []
    for rank in range(2):
        shard, first = D.shard_bank(bank, 2, rank)
        det = api.Detector(ctx, 2, T)
        det.add_class(shard)
        det.finalize(w0, h0)
        det.match_quantized(qs, 70.0)
        buf = torch.empty(k * MATCH_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
        det.export_topk(0, k, first, buf.data_ptr())
        ctx.synchronize()
        bufs.append(buf.cpu().numpy().view(MATCH_DTYPE).copy())
        det.close()
    merged = api.merge_topk(np.concatenate(bufs), k)
    full, n_full = oracle.match_quantized(qs, w0, h0, T, [bank], 70.0)
    assert n_full > 4 and len(merged) == min(k, n_full)
    assert merged.tobytes() == full[:len(merged)].tobytes()


def _run_bench(argv, timeout=900):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FL_DEV_POISON")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


SHARDED = ["--gpus", "2", "--shard", "templates", "--share-device", "--verify-sharded", "--templates", "30", "--batch", "6", "--scenes", "3",
           "--steps", "3", "--warmup", "1", "--icp-iters", "8", "--topk", "16"]


def test_template_sharded_recognition_two_ranks_on_one_gpu():
    """BASELINE configs[3] rehearsed on one GPU: two processes, each holding half of the bank, gloo in place of RCCL
    (two RCCL ranks cannot share a device), `bench.py --gpus 2 --shard templates --verify-sharded` STARTED PLAINLY (the parent
    launches the two ranks itself): every frame's best match and pose must equal, bit for bit, what one detector over the
    whole bank returns from fl_recognize_batch.  The device path runs: fl_export_topk_batch -> all-gather ->
    fl_select_best_batch -> fl_refine_selected -> int32 all-reduce of the pose rows, pipelined over three steps."""
    out = _run_bench(SHARDED)
    assert out["n_gpus"] == 2 and out["config"]["templates_total"] == 60
    assert out["verified_against_single_detector"] is True, out
    assert out["detections"] == "6/6"
    assert out["collectives"]["ranks"] == 2 and out["collectives"]["path"].startswith("device")
    assert min(out["winner_owner_histogram"]) > 0                   # both ranks own winners: both refined


def test_template_sharded_host_merge_path_gives_the_same(tmp_path):
    """the round-2 path (numpy merge on the host, fl_refine_matches) stays available and equal"""
    out = _run_bench(SHARDED + ["--sharded-host-merge"])
    assert out["verified_against_single_detector"] is True, out
    assert out["collectives"]["path"].startswith("host")


def test_template_sharded_overflow_is_seen_by_every_rank_and_grown():
    """Candidate buffers of 64 entries and a threshold that makes every non-zero coarse cell a candidate: every frame
    overflows on both ranks.  The exported records carry the flag through the all-gather, every rank grows its buffers
    (fl_detector_grow_candidates) and the step runs again; the result still equals the single detector's, and no truncated
    list is ever refined."""
    out = _run_bench(SHARDED + ["--max-candidates", "64", "--match-threshold", "-100"])
    assert out["verified_against_single_detector"] is True, out
    assert out["config"]["candidate_capacity"] > 64


def test_template_sharded_single_rank_line_has_roofline_and_cpu_baseline():
    out = _run_bench(["--shard", "templates", "--templates", "40", "--batch", "8", "--scenes", "3", "--steps", "3", "--warmup", "1",
                      "--icp-iters", "8", "--topk", "16", "--cpu-seconds", "2", "--compare-host-merge", "--verify-sharded"])
    assert out["n_gpus"] == 1 and out["verified_against_single_detector"] is True
    assert out["roofline"] is not None and out["roofline"]["launch_ms"] > 0 and out["cpu_baseline"]["value"] > 0
    assert out["collectives"]["host_syncs_per_step"] <= 1.5
    assert out["collectives"]["host_merge_comparison"]["same_result"] is True


def test_template_sharded_cxx_host_single_rank_line():
    """`bench.py --shard templates --mg-host cxx`: the step driven by the C++ host (libfealess_mg.so: ncclCommInitRank,
    ncclAllGather, ncclAllReduce from C++), one rank on the one GPU; every frame equals the single detector's result."""
    out = _run_bench(["--shard", "templates", "--mg-host", "cxx", "--templates", "40", "--batch", "8", "--scenes", "3", "--steps", "3",
                      "--warmup", "1", "--icp-iters", "8", "--topk", "16", "--no-cpu-baseline", "--verify-sharded", "--compare-host-merge"])
    assert out["n_gpus"] == 1 and out["verified_against_single_detector"] is True, out
    assert out["collectives"]["path"].startswith("C++ host")
    assert out["collectives"]["host_syncs_per_step"] <= 1.01
    assert out["collectives"]["host_merge_comparison"]["same_result"] is True
