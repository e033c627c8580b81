"""world_size-2 gloo tests of the N > 1 path (no GPU): template-sharded matching = per-rank match
on a contiguous slice of the bank + all-gather of fixed-size top-k records + exact merge, must
equal one Detector::match over the whole bank; and the frame-sharded split used by bench.py.
The per-rank matcher here is the oracle (tests may use it); on the GPU box the same host logic
runs with the HIP detector and backend "nccl" (RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_py as O
    from fealess_amd import synth
    from fealess_amd import distributed as D
    from fealess_amd.api import merge_topk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)                       # same frame + bank on every rank
    w0, h0, T = 320, 160, [5, 8]
    qs = [synth.random_quantized(rng, w0 >> l, h0 >> l, 0.04) for l in range(2) for _ in range(2)]
    bank = synth.make_bank("obj", 41, 2, 2, w0, h0, seed=17, qs=qs, planted_frac=0.5, bbox=64)
    shard, first = D.shard_bank(bank, world, rank)
    local, _ = O.match_quantized(qs, w0, h0, T, [shard], 55.0)
    gathered = D.allgather_records(D.pad_topk(local, k, first), dist)
    merged = merge_topk(gathered, k)
    full, n_full = O.match_quantized(qs, w0, h0, T, [bank], 55.0)
    ok = len(merged) == min(k, n_full) and merged.tobytes() == full[:len(merged)].tobytes() and n_full > 8
    # frame-sharded: disjoint, complete cover
    f0, fc = D.shard_range(37, world, rank)
    import torch
    cover = torch.zeros(37, dtype=torch.int32)
    cover[f0:f0 + fc] = 1
    dist.all_reduce(cover)
    ok = ok and bool((cover == 1).all())
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {len(merged)} {n_full}\n")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [8, 64])
def test_template_sharded_allgather_merge_equals_global(tmp_path, k):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, k, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, n_m, n_full = open(tmp_path / f"rank{r}.txt").read().split()
        assert ok == "1", (r, n_m, n_full)


def _worker_recognize(rank, world, port, out_dir):
    """template_sharded_recognize (the host logic bench.py --shard templates runs with RCCL + the HIP detector) with gloo and the
    oracle behind its four callables: per-rank match of the slice, all-gather, merge, refinement by the owner, pose sum."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_py as O
    from fealess_amd import synth
    from fealess_amd import distributed as D
    from fealess_amd.bank import MATCH_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, thr, k = [5, 8], 70.0, 16
    sc = synth.recognition_scene(lambda b, d, l: O.quantize_pyramid(b, d, l), levels=2, seed=13, n_views=5, n_random=6)
    bank, K = sc["bank"], sc["K"]
    # frame 1: the same scene shifted sideways
    frames = [(sc["bgr"], sc["depth"]), (np.roll(sc["bgr"], 14, axis=1), np.roll(sc["depth"], 14, axis=1))]
    n = bank.n_pyramids
    shard, first = D.shard_bank(bank, world, rank)

    def local_topk():
        out = np.zeros((len(frames), k), MATCH_DTYPE)
        for f, (b, d) in enumerate(frames):
            m, _ = O.match_images(b, d, T, [shard], thr)
            out[f] = D.pad_topk(m, k, first)
        return out

    def allgather(local):
        return D.allgather_records(local.reshape(-1), dist).reshape(world, len(frames), k)

    def refine(fr, matches):
        out = np.zeros((len(fr), 17), np.float32)
        for j, f in enumerate(fr):
            r = O.recognition(frames[f][0], frames[f][1], K, T, shard, thr, 8, 0.0, -3.0e38)
            # the global winner is its owner's local matches[0]
            assert r["best"]["template_id"] == int(matches["template_id"][j]) and r["best"]["x"] == int(matches["x"][j])
            out[j, 0] = r["found"]
            out[j, 1:] = r["pose"].reshape(-1)
        return out

    def allreduce_sum(a):
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t)
        return t.numpy()

    best, n_out, poses = D.template_sharded_recognize(len(frames), k, n, world, rank, local_topk, allgather, refine, allreduce_sum, full_lists=True)
    ok = True
    owners = []
    for f, (b, d) in enumerate(frames):
        e = O.recognition(b, d, K, T, bank, thr, 8, 0.0, -3.0e38)
        ok = ok and e["found"] == 1 and poses[f, 0] == 1.0
        ok = ok and int(best["template_id"][f]) == e["best"]["template_id"] and np.float32(best["similarity"][f]) == e["best"]["similarity"]
        ok = ok and np.array_equal(poses[f, 1:].view(np.uint32), e["pose"].reshape(-1).astype(np.float32).view(np.uint32))
        ok = ok and n_out[f] == min(k, e["n_matches"])
        owners.append(D.owner_of(int(best["template_id"][f]), n, world))
    # A candidate-buffer overflow on ONE rank (fl_export_topk_batch marks the frame's record 0 with TOPK_OVERFLOW): the flag
    # travels with the all-gather, so both ranks see it.  Without a grow callable the frame is reported as failed (no pose,
    # nothing refined from a truncated list); with one, every rank grows and the step runs again to the same result.
    state = {"grown": 0, "calls": 0}

    def local_topk_overflowing():
        out = local_topk()
        state["calls"] += 1
        if rank == 1 and not state["grown"]:
            out[1] = D.pad_topk(np.zeros(0, MATCH_DTYPE), k)
            out[1]["template_id"][0] = D.TOPK_OVERFLOW
        return out

    def refine_checked(fr, matches):
        assert (matches["template_id"] >= 0).all()
        return refine(fr, matches)

    b2, _, p2 = D.template_sharded_recognize(len(frames), k, n, world, rank, local_topk_overflowing, allgather, refine_checked, allreduce_sum)
    ok = ok and int(b2["template_id"][1]) == D.TOPK_OVERFLOW and p2[1, 0] == 0.0 and not p2[1].any()
    ok = ok and int(b2["template_id"][0]) == int(best["template_id"][0]) and np.array_equal(p2[0].view(np.uint32), poses[0].view(np.uint32))

    def grow():
        state["grown"] += 1

    state["calls"] = 0
    b3, _, p3 = D.template_sharded_recognize(len(frames), k, n, world, rank, local_topk_overflowing, allgather, refine_checked, allreduce_sum, grow=grow)
    ok = ok and state["grown"] == 1 and state["calls"] == 2          # both ranks grew once and ran the step twice
    ok = ok and b3.tobytes() == best.tobytes() and np.array_equal(p3.view(np.uint32), poses.view(np.uint32))
    # A rank whose buffers cannot grow (hard cap / out of memory: grow() raises on THAT rank only) must not leave the other
    # rank waiting in the next all-gather: the outcome of growing is all-reduced, both ranks stop retrying after the same
    # attempt and report the frame as TOPK_OVERFLOW.
    state.update(grown=0, calls=0)

    def grow_failing_on_rank1():
        if rank == 1:
            raise RuntimeError("fl_detector_grow_candidates: FL_ERR_OVERFLOW (hard cap)")

    b4, _, p4 = D.template_sharded_recognize(len(frames), k, n, world, rank, local_topk_overflowing, allgather, refine_checked, allreduce_sum,
                                             grow=grow_failing_on_rank1)
    ok = ok and state["calls"] == 1 and int(b4["template_id"][1]) == D.TOPK_OVERFLOW and not p4[1].any()
    ok = ok and int(b4["template_id"][0]) == int(best["template_id"][0]) and np.array_equal(p4[0].view(np.uint32), poses[0].view(np.uint32))
    # the pose rows travel as int32 bit patterns: a -0.0 of the owner survives the exchange (a float sum would give +0.0)
    z = np.zeros((1, 17), np.float32)
    if rank == 0:
        z[0, 3] = -0.0
        z[0, 0] = 1.0
    tot = allreduce_sum(z.view(np.int32)).view(np.float32)
    ok = ok and np.signbit(tot[0, 3]) and tot[0, 0] == 1.0
    with open(os.path.join(out_dir, f"reco{rank}.txt"), "w") as fh:
        fh.write(f"{int(ok)} {owners}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_template_sharded_recognition_with_icp_handoff(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_recognize, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok = open(tmp_path / f"reco{r}.txt").read().split()[0]
        assert ok == "1", open(tmp_path / f"reco{r}.txt").read()


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from fealess_amd.distributed import shard_range
    for n in (0, 1, 7, 16000, 2001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a, ca), (b, _) in zip(spans, spans[1:]):
                assert a + ca == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_best_of_ranks_equals_first_of_the_merged_list():
    """The fast path of template_sharded_recognize: matches[0] of the global sort = the best of the ranks' first records,
    against fl_merge_topk_batch on random sorted per-rank lists (ties in similarity and template id included)."""
    sys.path.insert(0, ROOT)
    from fealess_amd import distributed as D
    from fealess_amd.api import merge_topk_batch
    from fealess_amd.bank import MATCH_DTYPE
    rng = np.random.default_rng(5)
    world, n_frames, k, n_templates = 4, 50, 8, 103
    g = np.zeros((world, n_frames, k), MATCH_DTYPE)
    g["template_id"] = -1
    g["class_idx"] = -1
    for r in range(world):
        first, count = D.shard_range(n_templates, world, r)
        for f in range(n_frames):
            n = int(rng.integers(0, k + 1))
            rec = np.zeros(n, MATCH_DTYPE)
            rec["similarity"] = rng.choice(np.array([80.0, 85.5, 90.25, 97.0], np.float32), n)   # few values: many ties
            rec["template_id"] = first + rng.integers(0, count, n)
            rec["x"], rec["y"] = rng.integers(0, 640, n), rng.integers(0, 480, n)
            rec = rec[np.lexsort((rec["x"], rec["y"], rec["class_idx"], rec["template_id"], -rec["similarity"]))]
            g[r, f, :n] = rec
    best = D.best_of_ranks(g)
    merged, n_out = merge_topk_batch(g.reshape(-1), world, n_frames, k, k)
    for f in range(n_frames):
        if n_out[f] == 0:
            assert best["template_id"][f] < 0
        else:
            assert best[f] == merged[f, 0], f
    own = D.owners_of(best["template_id"], n_templates, world)
    for f in range(n_frames):
        assert own[f] == (D.owner_of(int(best["template_id"][f]), n_templates, world) if best["template_id"][f] >= 0 else -1)
