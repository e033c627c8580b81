"""bench.py --shard templates: BASELINE configs[3] (16000 templates over 8 GPUs = 2000 per GPU) end to end.

Every rank holds `--templates` templates of a bank of world * templates (a contiguous slice, with its depth renders),
all ranks see the same frames.  One step = per rank: front-end + Detector::match of its slice (fl_match_batch_submit),
top-k export of every frame in one launch (fl_export_topk_batch); all-gather of the records (RCCL,
all_gather_into_tensor: world * frames * k * 20 bytes); merge on every rank as one Detector::match over the whole bank
would order them; the rank that owns a frame's winning template refines it (fl_refine_matches = the second half of
Recognition(), obj_reco_lmicp.cpp:111-197); the poses are summed over the ranks (all_reduce).  The library's stream is
the torch stream, so the collectives are ordered after the kernels without a host synchronisation in between.

`--verify-sharded` (rehearsals, tests): rank 0 also builds a single detector over the whole bank and checks that every
frame's best match and pose equal fl_recognize_batch's, bit for bit.
"""
import os
import time

import numpy as np

from . import distributed as D
from .bank import MATCH_DTYPE

FORCE_ALL_ITERS = -3.0e38


def run(args, ctx, dist, world, rank, w, h, K, build_bank, build_frames):
    import torch
    from . import api
    from . import _lib as L
    device = torch.device("cuda", ctx.device)
    n_total = args.templates * world
    bank, scenes = build_bank(ctx, args, n_total, w, h, K)                 # same bank on every rank ...
    first, count = D.shard_range(n_total, world, rank)
    shard = bank.subset(first, count)                                      # ... of which this rank keeps its slice
    B = args.batch
    bgrs, depths = build_frames(scenes, B, rank, w, h, same_on_all_ranks=True)
    T = [5, 8, 4][:args.levels] if args.levels == 3 else [5, 8][:args.levels]
    det = api.Detector(ctx, 2, T)
    det.add_class(shard)
    det.finalize(w, h, max_batch=B, max_candidates=4096)
    # kernels and collectives on one timeline: an explicit torch stream (the default stream's handle is 0, which
    # fl_context_set_stream reads as "back to the context's own stream")
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    d_bgr = torch.from_numpy(bgrs).to(device)
    d_depth = torch.from_numpy(depths.view(np.int16)).to(device)
    bptr = [d_bgr.data_ptr() + i * w * h * 3 for i in range(B)]
    dptr = [d_depth.data_ptr() + i * w * h * 2 for i in range(B)]
    mode = {"parity": L.FL_ICP_PARITY, "fast": L.FL_ICP_FAST, "plane": L.FL_ICP_POINT_TO_PLANE}[args.icp_mode]
    params = L.RecognitionParams(75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, mode)
    k = args.topk
    rec_bytes = B * k * MATCH_DTYPE.itemsize
    local = torch.empty(rec_bytes, dtype=torch.uint8, device=device)
    gathered = torch.empty(world * rec_bytes, dtype=torch.uint8, device=device)
    on_gpu_collectives = dist is not None and dist.get_backend() == "nccl"

    def local_topk():
        det.match_batch_submit(bptr, dptr, 75.0)
        det.export_topk_batch(B, k, first, local.data_ptr())
        return local

    def allgather(t):
        if dist is None:
            return t.cpu().numpy().view(MATCH_DTYPE).reshape(1, B, k)
        if on_gpu_collectives:
            dist.all_gather_into_tensor(gathered, t)
            return gathered.cpu().numpy().view(MATCH_DTYPE).reshape(world, B, k)
        tc = t.cpu()                                                        # gloo rehearsal: staged through the host
        out = torch.empty(world * rec_bytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, tc)
        return out.numpy().view(MATCH_DTYPE).reshape(world, B, k)

    def refine(frames, matches):
        res = det.refine_matches(frames, matches, K, params)
        a = np.frombuffer(res, dtype=api.RESULT_DTYPE)              # no per-record Python work: 2048 jobs per step
        out = np.empty((len(a), 17), np.float32)
        out[:, 0] = a["found"]
        out[:, 1:] = a["pose"]
        return out

    def allreduce_sum(a):
        if dist is None:
            return a
        t = torch.from_numpy(a)
        if on_gpu_collectives:
            t = t.to(device)
        dist.all_reduce(t)
        return t.cpu().numpy()

    def step():
        return D.template_sharded_recognize(B, k, n_total, world, rank, local_topk, allgather, refine, allreduce_sum)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best, n_out, poses = step()
    sync_all()
    el = time.perf_counter() - t0
    if dist is not None:
        tm = torch.tensor([el], dtype=torch.float64, device=device if on_gpu_collectives else "cpu")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        el = float(tm.item())
    verified = None
    mismatches = []
    if getattr(args, "verify_sharded", False) and rank == 0:
        full = api.Detector(ctx, 2, T)
        full.add_class(bank)
        full.finalize(w, h, max_batch=B, max_candidates=4096)
        full.recognize_submit_device(bptr, dptr, K, params)
        ref = full.recognize_collect(B)
        verified = True
        mismatches = []
        for f in range(B):
            r = ref[f]
            got = dict(found=int(poses[f, 0]), tid=int(best["template_id"][f]), x=int(best["x"][f]), y=int(best["y"][f]),
                       sim=float(best["similarity"][f]))
            exp = dict(found=int(r.found), tid=int(r.best.template_id), x=int(r.best.x), y=int(r.best.y), sim=float(np.float32(r.best.similarity)))
            same = got["found"] == exp["found"]
            if r.found:
                same = same and got == exp and np.array_equal(np.array(list(r.pose), np.float32).view(np.uint32), poses[f, 1:].view(np.uint32))
            if not same:
                mismatches.append(dict(frame=f, got=got, exp=exp, pose_maxdiff=float(np.abs(np.array(list(r.pose), np.float32) - poses[f, 1:]).max())))
            verified = verified and bool(same)
        full.close()
    found = int((poses[:, 0] > 0).sum())
    own = D.owners_of(best["template_id"], n_total, world)
    owners = np.bincount(own[own >= 0], minlength=world).tolist()
    det.close()
    return {
        "metric": "frames/sec (640x480 RGB-D x N templates, 20 ICP iters)",
        "value": round(B * args.steps / el, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {w}x{h}, {n_total} templates sharded {args.templates} per GPU over {world} GPUs, "
                               f"{args.levels} pyramid levels T={T}, top-{k} all-gather + merge, winner refined by its owner "
                               f"({args.icp_iters} ICP iterations forced), ICP mode {args.icp_mode}",
                   "frames_per_step": B, "templates_total": n_total, "templates_per_gpu": args.templates, "levels": args.levels,
                   "parallelism": f"template-sharded x{world}",
                   "note": "weak scaling in TEMPLATES: the frames are the same on every rank, the bank grows with the ranks"},
        "collectives": {"backend": (dist.get_backend() if dist is not None else None), "ranks": world,
                        "all_gather_bytes_per_step": world * rec_bytes, "all_reduce_bytes_per_step": B * 17 * 4},
        "detections": f"{found}/{B}", "winner_owner_histogram": owners, "verified_against_single_detector": verified,
        "verify_mismatches": (mismatches[:4] if verified is False else None),
        "roofline": None, "cpu_baseline": None,
    }
