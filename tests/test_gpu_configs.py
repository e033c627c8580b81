"""BASELINE.json configs[3] and configs[4] at their REAL sizes on the one GPU a test box has.

configs[3]: 640x480, 16000 templates sharded 2000 per GPU, all-gather of the per-GPU top-k detections.  Here the eight
"ranks" are eight detectors on one device, each holding its contiguous 2000-template slice of ONE 16000-pyramid bank;
their fl_export_topk_batch buffers are laid side by side exactly as ncclAllGather would leave them, every rank runs
fl_select_best_batch / fl_refine_selected on that buffer, and the int32 sum of the rows is the result.  It must equal
one 16000-template detector bit for bit and, on a sample of frames, the oracle (the N-template loop linemod.cpp:1458, the
global std::sort + std::unique :1437-1439, matches[0] obj_reco_lmicp.cpp:111).

configs[4]: 64 frames x 2000 templates, frame-sharded 8 per GPU: eight batches of 8 = one batch of 64 = the oracle on a
sample of the frames.
"""
import numpy as np
import pytest

from fealess_amd import _lib as L
from fealess_amd import api, synth
from fealess_amd.bank import MATCH_DTYPE, TemplateBank

pytestmark = pytest.mark.gpu

T2 = [5, 8]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _pyramid_of(bank, j):
    """(templates, pose13, depth render) of pyramid j of `bank`, ready for TemplateBank.add_pyramid"""
    v = bank.subset(j, 1)
    t, f, p = v.arrays()
    tl = []
    for k in range(len(t)):
        h = t[k]
        fr = f[h["feat_begin"]:h["feat_begin"] + h["feat_count"]]
        tl.append(dict(width=int(h["width"]), height=int(h["height"]), offset_x=int(h["offset_x"]), offset_y=int(h["offset_y"]),
                       pyramid_level=int(h["pyramid_level"]), features=np.stack([fr["x"], fr["y"], fr["label"]], axis=1)))
    return tl, p[0], (v.model_depths[0] if v.model_depths else None)


def _big_bank(oracle, n_total, trained_at, seed):
    """One class of n_total pyramids and len(trained_at) scenes of the same object in different poses: the near-true rendered
    view of scene j (with its depth render and pose) sits at id trained_at[j], random pyramids everywhere else."""
    scenes = [synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=seed + 7 * j, n_views=1)
              for j in range(len(trained_at))]
    assert all(sc["bank"].n_pyramids == 1 for sc in scenes)
    rng = np.random.default_rng(seed + 77)
    bank = TemplateBank("obj", 2, 2)
    at = {tid: j for j, tid in enumerate(trained_at)}
    for i in range(n_total):
        if i in at:
            bank.add_pyramid(*_pyramid_of(scenes[at[i]]["bank"], 0))
        else:
            bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, None)
    return scenes, bank


def test_c4_16000_templates_sharded_2000_per_rank_equals_one_detector_and_the_oracle(ctx, oracle):
    import torch
    n_total, world, per, k = 16000, 8, 2000, 64
    # four scenes; the view that wins scene 0 lives in the LAST shard (global id 14000 + 1234: template_id_base = 14000 is
    # exercised by a winner), the others in shards 0, 2 and 5; a fifth view sits at the very last id
    trained_at = [15234, 17, 4100, 11999, 15999]
    scenes, bank = _big_bank(oracle, n_total, trained_at, seed=31)
    sc = scenes[0]
    frames_b = [s_["bgr"] for s_ in scenes[:4]] + [np.roll(scenes[0]["bgr"], 8, axis=1), np.roll(scenes[2]["bgr"], -14, axis=1)]
    frames_d = [s_["depth"] for s_ in scenes[:4]] + [np.roll(scenes[0]["depth"], 8, axis=1), np.roll(scenes[2]["depth"], -14, axis=1)]
    shifts = list(range(len(frames_b)))
    n = len(shifts)
    d_b = torch.from_numpy(np.stack(frames_b)).cuda()
    d_d = torch.from_numpy(np.stack(frames_d).view(np.int16)).cuda()
    torch.cuda.synchronize()
    bp = [d_b.data_ptr() + i * 640 * 480 * 3 for i in range(n)]
    dp = [d_d.data_ptr() + i * 640 * 480 * 2 for i in range(n)]
    params = L.RecognitionParams(75.0, 10, 0.5, 0.01, L.FL_ICP_PARITY)

    # ONE detector over the whole bank
    full = api.Detector(ctx, 2, T2)
    full.add_class(bank)
    full.finalize(640, 480, max_batch=n)
    assert full.num_templates() == n_total
    full.recognize_submit_device(bp, dp, sc["K"], params)
    ref = [api.recognition_result_to_dict(r) for r in full.recognize_collect(n)]
    full.close()
    assert all(r["status"] == 0 and r["found"] == 1 for r in ref)
    assert {r["best"]["template_id"] for r in ref} <= set(trained_at)
    assert any(r["best"]["template_id"] >= 14000 for r in ref)       # a winner behind template_id_base = 14000
    assert len({r["best"]["template_id"] // per for r in ref}) >= 3  # winners (and ICP work) on several ranks

    # eight ranks of 2000 on this device
    rec = MATCH_DTYPE.itemsize
    gathered = torch.zeros(world * n * k * rec, dtype=torch.uint8, device="cuda")      # [rank][frame][k] records: ncclAllGather's layout
    dets = []
    for r in range(world):
        det = api.Detector(ctx, 2, T2)
        det.add_class(bank.subset(r * per, per))
        det.finalize(640, 480, max_batch=n)
        det.match_batch_submit(bp, dp, 75.0)
        det.export_topk_batch(n, k, r * per, gathered.data_ptr() + r * n * k * rec)
        dets.append(det)
    ctx.synchronize()
    g = gathered.cpu().numpy().view(MATCH_DTYPE).reshape(world, n, k)
    assert int(g["template_id"].max()) >= 14000 and int(g["template_id"].max()) < n_total
    for r in range(world):                                               # global ids: every rank's records lie in its slice
        ids = g[r]["template_id"]
        assert ((ids == -1) | ((ids >= r * per) & (ids < (r + 1) * per))).all()
    rows = torch.zeros((world, n, 17), dtype=torch.float32, device="cuda")
    best = torch.zeros((world, n * rec), dtype=torch.uint8, device="cuda")
    for r, det in enumerate(dets):
        det.select_best_batch(gathered.data_ptr(), world, n, k, r * per, per, best[r].data_ptr())
        det.refine_selected(n, sc["K"], params, rows[r].data_ptr())
    ctx.synchronize()
    torch.cuda.synchronize()
    best_h = best.cpu().numpy().view(MATCH_DTYPE).reshape(world, n)
    for r in range(1, world):
        assert best_h[r].tobytes() == best_h[0].tobytes()               # every rank selects the same winners
    total = rows.cpu().numpy().view(np.int32).sum(axis=0, dtype=np.int64).astype(np.int32).view(np.float32)   # the all-reduce of bit patterns
    owners = set()
    for f in range(n):
        e = ref[f]
        b = best_h[0][f]
        assert (int(b["x"]), int(b["y"]), int(b["template_id"]), int(b["class_idx"])) == (e["best"]["x"], e["best"]["y"], e["best"]["template_id"], 0), f
        assert np.float32(b["similarity"]) == e["best"]["similarity"]
        assert total[f, 0] == 1.0 and np.array_equal(_bits(total[f, 1:]), _bits(e["pose"].reshape(-1))), f
        owners.add(int(b["template_id"]) // per)
        assert sum(float(rows[r, f, 0]) for r in range(world)) == 1.0   # exactly one owner refined the frame
    for det in dets:
        det.close()
    # and the oracle on two of the frames (Detector::match over all 16000 templates + the refinement of matches[0])
    for f in (0, 1, 5):
        e = oracle.recognition(frames_b[f], frames_d[f], sc["K"], T2, bank, 75.0, 10, 0.5, 0.01)
        r = ref[f]
        assert e["found"] == 1 and r["n_matches"] == e["n_matches"]
        assert r["best"]["template_id"] == e["best"]["template_id"] and r["best"]["x"] == e["best"]["x"] and r["best"]["y"] == e["best"]["y"]
        assert r["best"]["similarity"] == e["best"]["similarity"]
        assert r["det"]["n_points"] == e["det"]["n_points"] and np.array_equal(_bits(r["pose"]), _bits(e["pose"]))
    print("c4: winners", sorted({r["best"]["template_id"] for r in ref}), "owner ranks", sorted(owners))


def test_c5_64_frames_x_2000_templates_in_batches_of_8_equal_one_batch_and_the_oracle(ctx, oracle):
    scenes = [synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=s, n_views=3) for s in (3, 5)]
    # one 2000-pyramid bank holding both scenes' views (ids 0-2 and 1000-1002) among random pyramids
    rng = np.random.default_rng(9)
    bank = TemplateBank("obj", 2, 2)
    for i in range(2000):
        src = scenes[0]["bank"] if i < 3 else (scenes[1]["bank"] if 1000 <= i < 1003 else None)
        if src is None:
            bank.add_pyramid(synth.random_pyramid(rng, 2, 2, 640, 480), None, None)
            continue
        bank.add_pyramid(*_pyramid_of(src, i % 1000))
    K = scenes[0]["K"]
    frames_b, frames_d = [], []
    for i in range(64):                                         # 64 distinct frames: two scenes x 32 sideways shifts
        s = scenes[i % 2]
        sh = 2 * (i // 2) - 30
        frames_b.append(np.roll(s["bgr"], sh, axis=1))
        frames_d.append(np.roll(s["depth"], sh, axis=1))
    det = api.Detector(ctx, 2, T2)
    det.add_class(bank)
    det.finalize(640, 480, max_batch=64)
    whole = det.recognize_batch(frames_b, frames_d, K, 75.0, 20, -1.0, -3.0e38)      # the bench's 20 forced iterations
    parts = []
    for r in range(8):                                          # configs[4]'s per-GPU shape: 8 frames x 2000 templates
        parts += det.recognize_batch(frames_b[8 * r:8 * r + 8], frames_d[8 * r:8 * r + 8], K, 75.0, 20, -1.0, -3.0e38)
    det.close()
    assert sum(r["found"] for r in whole) >= 60
    for i, (a, b) in enumerate(zip(whole, parts)):
        assert a["status"] == b["status"] == 0 and a["found"] == b["found"] and a["n_matches"] == b["n_matches"] and a["best"] == b["best"], i
        assert a["det"]["n_points"] == b["det"]["n_points"] and np.array_equal(_bits(a["pose"]), _bits(b["pose"])), i
        assert a["det"]["icp"]["iters"] == b["det"]["icp"]["iters"] and (not a["found"] or a["det"]["icp"]["iters"] == 20)
    for i in (0, 9, 34, 63):                                    # the oracle on a sample (both scenes, both ends of the shifts)
        e = oracle.recognition(frames_b[i], frames_d[i], K, T2, bank, 75.0, 20, -1.0, -3.0e38)
        g = whole[i]
        assert g["found"] == e["found"] and g["n_matches"] == e["n_matches"], i
        if e["found"]:
            assert g["best"]["template_id"] == e["best"]["template_id"] and g["best"]["x"] == e["best"]["x"] and g["best"]["y"] == e["best"]["y"]
            assert g["best"]["similarity"] == e["best"]["similarity"]
            assert g["det"]["n_points"] == e["det"]["n_points"] and np.array_equal(_bits(g["pose"]), _bits(e["pose"])), i


def test_cxx_multi_gpu_host_with_one_rank_equals_fl_recognize_batch(ctx, oracle):
    """libfealess_mg.so (include/fealess_mg.h): the C++ host of the template-sharded path, RCCL initialised with ONE rank
    (all-gather of 1, all-reduce of 1: the collectives run, on the context's stream) = fl_recognize_batch on the same bank,
    bit for bit; also with 64-entry candidate buffers that every frame overflows (grown inside fl_mg_recognize_batch)."""
    import torch
    sc = synth.recognition_scene(lambda b, d, l: oracle.quantize_pyramid(b, d, l), levels=2, seed=13, n_views=4, n_random=56)
    frames_b = [sc["bgr"], np.roll(sc["bgr"], 10, axis=1), np.full_like(sc["bgr"], 90), np.roll(sc["bgr"], -16, axis=1)]
    frames_d = [sc["depth"], np.roll(sc["depth"], 10, axis=1), np.full_like(sc["depth"], 1200), np.roll(sc["depth"], -16, axis=1)]
    n = len(frames_b)
    d_b = torch.from_numpy(np.stack(frames_b)).cuda()
    d_d = torch.from_numpy(np.stack(frames_d).view(np.int16)).cuda()
    torch.cuda.synchronize()
    bp = [d_b.data_ptr() + i * 640 * 480 * 3 for i in range(n)]
    dp = [d_d.data_ptr() + i * 640 * 480 * 2 for i in range(n)]
    ref_det = api.Detector(ctx, 2, T2)
    ref_det.add_class(sc["bank"])
    ref_det.finalize(640, 480, max_batch=n)
    for thr, cap in ((75.0, 0), (-100.0, 64)):
        params = L.RecognitionParams(thr, 10, 0.5, 0.01, L.FL_ICP_PARITY)
        ref = ref_det.recognize_batch(frames_b, frames_d, sc["K"], thr, 10, 0.5, 0.01)
        det = api.Detector(ctx, 2, T2)
        det.add_class(sc["bank"])
        det.finalize(640, 480, max_batch=n, max_candidates=cap)
        mg = api.MgGroup(det, api.MgGroup.unique_id(), 1, 0, 0, sc["bank"].n_pyramids, 16)
        got = mg.recognize_batch(bp, dp, sc["K"], params)
        st = mg.stats()
        assert st["allgather_bytes"] == n * 16 * 20 and st["allreduce_bytes"] == n * 17 * 4
        assert (st["attempts"] > 1) == (cap == 64)
        for f in range(n):
            e, g = ref[f], got[f]
            assert g.status == 0 and g.found == e["found"], (thr, f)
            if e["n_matches"] > 0:
                assert (g.best.x, g.best.y, g.best.template_id, g.best.class_idx) == (e["best"]["x"], e["best"]["y"], e["best"]["template_id"], e["best"]["class_idx"])
                assert np.float32(g.best.similarity) == e["best"]["similarity"]
            else:
                assert g.best.template_id == -1
            if e["found"]:
                assert np.array_equal(_bits(np.array(g.pose, np.float32)), _bits(e["pose"].reshape(-1))), (thr, f)
        if cap == 0:
            assert ref[0]["found"] == 1 and ref[2]["found"] == 0     # (at threshold 75: the blank frame matches nothing)
        mg.close()
        det.close()
    ref_det.close()


def test_context_options_are_explicit_state_not_the_environment(ctx, monkeypatch):
    """fl_context_set_option / _get_option: the development switches are context state.  Their initial values come from the
    environment ONCE, at fl_context_create; a variable that appears in the process environment afterwards changes nothing
    (round 3 read FL_SCAN_PRUNE & co. with getenv on every launch)."""
    assert ctx.get_option("scan_prune") == 1 and ctx.get_option("icp_wide") == -1 and ctx.get_option("icp_order") == 1
    monkeypatch.setenv("FL_SCAN_PRUNE", "0")                 # a stray variable in a host's environment, after the context exists
    monkeypatch.setenv("FL_ICP_WIDE", "1")
    assert ctx.get_option("scan_prune") == 1 and ctx.get_option("icp_wide") == -1
    c2 = api.Context(0)                                       # a context created NOW takes them as its initial values
    assert c2.get_option("scan_prune") == 0 and c2.get_option("icp_wide") == 1
    c2.set_option("scan_prune", 1)
    assert c2.get_option("scan_prune") == 1
    with pytest.raises(api.FealessError):
        c2.set_option("no_such_switch", 1)
    c2.close()
