"""bench.py --shard templates: BASELINE configs[3] (16000 templates over 8 GPUs = 2000 per GPU) end to end.

Every rank holds `--templates` templates of a bank of world * templates (a contiguous slice, with its depth renders),
all ranks see the same frames.  One step, per rank, with nothing but device work between Detector::match and the pose:

  compute stream   fl_match_batch_submit (front-end + match of the slice) -> fl_export_topk_batch (k records per frame)
  comm stream      all_gather_into_tensor of the records (RCCL; world * frames * k * 20 bytes, latency-bound over xGMI)
  compute stream   fl_select_best_batch: per frame matches[0] of one global std::sort + std::unique over all ranks' lists
                   (linemod.cpp:1437-1439; Recognition() uses nothing else, obj_reco_lmicp.cpp:111) and the jobs of the
                   frames whose winner this rank owns -> fl_refine_selected (the second half of Recognition(),
                   obj_reco_lmicp.cpp:111-197, one launch) -> {found, pose} rows
  comm stream      all_reduce(SUM) of the rows as int32 bit patterns (one owner per frame, zeros elsewhere: exact)
  host             one wait per step, for the rows of a step queued earlier

The steps are software-pipelined: the match of step i+1 is queued before the select / refine of step i, so the all-gather
of step i runs underneath the scan of step i+1 on the second stream.  `--sharded-host-merge` runs the round-2 path instead
(records to the host, numpy merge, fl_refine_matches: three host synchronisations per step) for the before/after figure;
a gloo rehearsal (`--share-device`) runs the same device kernels but has to stage its two collectives through the host
(three synchronisations per step).

`--verify-sharded` (rehearsals, tests): rank 0 also builds a single detector over the whole bank and checks that every
frame's best match and pose equal fl_recognize_batch's, bit for bit.
"""
import time

import numpy as np

from . import distributed as D
from .bank import MATCH_DTYPE

FORCE_ALL_ITERS = -3.0e38
HBM_PEAK_GBS = 8000.0


def run(args, ctx, dist, world, rank, w, h, K, build_bank, build_frames, cpu_baseline=None):
    import torch
    from . import api
    from . import _lib as L
    device = torch.device("cuda", ctx.device)
    n_total = args.templates * world
    bank, scenes = build_bank(ctx, args, n_total, w, h, K, spread_trained=True)   # same bank on every rank ...
    first, count = D.shard_range(n_total, world, rank)
    shard = bank.subset(first, count)                                      # ... of which this rank keeps its slice
    B = args.batch
    bgrs, depths = build_frames(scenes, B, rank, w, h, same_on_all_ranks=True)
    T = [5, 8, 4][:args.levels] if args.levels == 3 else [5, 8][:args.levels]
    det = api.Detector(ctx, 2, T)
    det.add_class(shard)
    det.finalize(w, h, max_batch=B, max_candidates=args.max_candidates)
    # kernels and collectives on explicit torch streams (the default stream's handle is 0, which fl_context_set_stream
    # reads as "back to the context's own stream")
    compute = torch.cuda.Stream(device)
    comm = torch.cuda.Stream(device)
    torch.cuda.set_stream(compute)
    ctx.set_stream(compute.cuda_stream)
    d_bgr = torch.from_numpy(bgrs).to(device)
    d_depth = torch.from_numpy(depths.view(np.int16)).to(device)
    bptr = [d_bgr.data_ptr() + i * w * h * 3 for i in range(B)]
    dptr = [d_depth.data_ptr() + i * w * h * 2 for i in range(B)]
    mode = {"parity": L.FL_ICP_PARITY, "fast": L.FL_ICP_FAST, "plane": L.FL_ICP_POINT_TO_PLANE}[args.icp_mode]
    thr = float(args.match_threshold)
    params = L.RecognitionParams(thr, args.icp_iters, -1.0, FORCE_ALL_ITERS, mode)
    k = args.topk
    rec = MATCH_DTYPE.itemsize
    rec_bytes = B * k * rec
    on_gpu_collectives = dist is not None and dist.get_backend() == "nccl"
    device_path = not args.sharded_host_merge

    # ---- the device path: two slots of every buffer, step i uses slot i % 2 -------------------------------------------
    local = [torch.empty(rec_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
    gathered = [torch.empty(world * rec_bytes, dtype=torch.uint8, device=device) for _ in range(2)] if dist is not None else local
    best_d = [torch.empty(B * rec, dtype=torch.uint8, device=device) for _ in range(2)]
    rows_d = [torch.zeros(B * 17, dtype=torch.float32, device=device) for _ in range(2)]
    best_h = [torch.empty(B * rec, dtype=torch.uint8).pin_memory() for _ in range(2)]
    rows_h = [torch.empty(B * 17, dtype=torch.float32).pin_memory() for _ in range(2)]
    ev_exp = [torch.cuda.Event() for _ in range(2)]
    ev_gat = [torch.cuda.Event() for _ in range(2)]
    ev_ref = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_icp = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    host_syncs = [0]

    staged = dist is not None and not on_gpu_collectives      # gloo rehearsal: the collectives go through host memory

    def queue_match(s):
        det.match_batch_submit(bptr, dptr, thr)
        det.export_topk_batch(B, k, first, local[s].data_ptr())
        ev_exp[s].record(compute)
        if staged:
            ev_exp[s].synchronize()
            host_syncs[0] += 1
            out = torch.empty(world * rec_bytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, local[s].cpu())
            gathered[s].copy_(out)
            ev_gat[s].record(compute)
        elif dist is not None:
            with torch.cuda.stream(comm):
                comm.wait_event(ev_exp[s])
                dist.all_gather_into_tensor(gathered[s], local[s])
                ev_gat[s].record(comm)

    def queue_refine(s):
        if dist is not None:
            compute.wait_event(ev_gat[s])
        det.select_best_batch(gathered[s].data_ptr(), world, B, k, first, count, best_d[s].data_ptr())
        ev_icp[0].record(compute)
        det.refine_selected(B, K, params, rows_d[s].data_ptr(), depth_base=dptr[0], depth_stride=w * h * 2)
        ev_icp[1].record(compute)
        ev_ref[s].record(compute)
        if staged:
            ev_ref[s].synchronize()
            host_syncs[0] += 1
            t = rows_d[s].view(torch.int32).cpu()
            dist.all_reduce(t)                                                # one owner per frame: the int32 sum is exact
            rows_d[s].view(torch.int32).copy_(t)
        with torch.cuda.stream(comm):
            comm.wait_event(ev_ref[s])
            if staged:
                comm.wait_stream(compute)
            elif dist is not None:
                dist.all_reduce(rows_d[s].view(torch.int32))                  # one owner per frame: the int32 sum is exact
            best_h[s].copy_(best_d[s], non_blocking=True)
            rows_h[s].copy_(rows_d[s], non_blocking=True)
            ev_done[s].record(comm)

    def wait_results(s):
        ev_done[s].synchronize()
        host_syncs[0] += 1
        return best_h[s].numpy().view(MATCH_DTYPE).copy(), rows_h[s].numpy().reshape(B, 17).copy()

    def steps_device(n):
        """n pipelined steps; returns the last step's (best, poses)."""
        out = None
        for i in range(n):
            queue_match(i % 2)                                                  # match of step i ...
            if i > 0:
                queue_refine((i - 1) % 2)                                       # ... queued ahead of the refinement of step i - 1
            if i > 1:
                out = wait_results(i % 2)                                       # step i - 2 (same slot as i): the one host wait of this step
        if n > 0:
            queue_refine((n - 1) % 2)
        if n > 1:
            out = wait_results((n - 2) % 2)
        if n > 0:
            out = wait_results((n - 1) % 2)
        return out

    # ---- the host-merge path (round 2; also what a gloo rehearsal can do) ---------------------------------------------
    def local_topk():
        det.match_batch_submit(bptr, dptr, thr)
        det.export_topk_batch(B, k, first, local[0].data_ptr())
        return local[0]

    def allgather(t):
        host_syncs[0] += 1
        if dist is None:
            return t.cpu().numpy().view(MATCH_DTYPE).reshape(1, B, k)
        if on_gpu_collectives:
            dist.all_gather_into_tensor(gathered[0], t)
            return gathered[0].cpu().numpy().view(MATCH_DTYPE).reshape(world, B, k)
        tc = t.cpu()                                                        # gloo rehearsal: staged through the host
        out = torch.empty(world * rec_bytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, tc)
        return out.numpy().view(MATCH_DTYPE).reshape(world, B, k)

    def refine(frames, matches):
        host_syncs[0] += 1
        res = det.refine_matches(frames, matches, K, params)
        a = np.frombuffer(res, dtype=api.RESULT_DTYPE)              # no per-record Python work: 2048 jobs per step
        out = np.empty((len(a), 17), np.float32)
        out[:, 0] = a["found"]
        out[:, 1:] = a["pose"]
        return out

    def allreduce_sum(a):
        if dist is None:
            return a
        t = torch.from_numpy(np.ascontiguousarray(a))
        if on_gpu_collectives:
            host_syncs[0] += 1
            t = t.to(device)
        dist.all_reduce(t)
        return t.cpu().numpy()

    def grow():
        det.grow_candidates(B)

    def step_host():
        return D.template_sharded_recognize(B, k, n_total, world, rank, local_topk, allgather, refine, allreduce_sum, grow=grow)

    def steps_host(n):
        out = None
        for _ in range(n):
            b, _n, p = step_host()
            out = (b, p)
        return out

    # ---- the C++ host (libfealess_mg.so, include/fealess_mg.h): the same step driven from C++ directly on RCCL ----------
    cxx = getattr(args, "mg_host", "python") == "cxx"
    mg = None
    if cxx:
        if dist is not None and not on_gpu_collectives:
            raise SystemExit("--mg-host cxx drives RCCL itself: one GPU per rank (no --share-device / gloo rehearsal)")
        uid = [api.MgGroup.unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)                          # ncclGetUniqueId on rank 0, handed to every rank
        mg = api.MgGroup(det, uid[0], world, rank, first, count, k)
        MG_DT = np.dtype([("status", "<i4"), ("found", "<i4"), ("best", MATCH_DTYPE), ("pose", "<f4", 16)])

        def steps_cxx(n):
            out = None
            for _ in range(n):
                res = mg.recognize_batch(bptr, dptr, K, params)
                host_syncs[0] += mg.stats()["attempts"]                     # one stream synchronisation per attempt
                a = np.frombuffer(res, dtype=MG_DT)
                b = a["best"].copy()
                b["template_id"][a["status"] != 0] = D.TOPK_OVERFLOW
                p = np.zeros((B, 17), np.float32)
                p[:, 0] = a["found"]
                p[:, 1:] = a["pose"]
                out = (b, p)
            return out

    def sync_all():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(fn, steps, warmup):
        fn(warmup)
        sync_all()
        host_syncs[0] = 0
        t0 = time.perf_counter()
        out = fn(steps)
        sync_all()
        el = time.perf_counter() - t0
        syncs = host_syncs[0] / max(1, steps)
        if dist is not None:
            tm = torch.tensor([el], dtype=torch.float64, device=device if on_gpu_collectives else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        return el, out, syncs

    # one unpipelined step first: it grows the candidate buffers if a frame needs it (every rank sees the flag in the gathered
    # records and takes the same branch), so the timed steps run with buffers that hold every frame's list
    best0, _n0, poses0 = step_host()
    if (best0["template_id"] == D.TOPK_OVERFLOW).any():
        raise RuntimeError("a frame's candidate buffers still overflow after growing them")
    sync_all()
    main_fn = steps_cxx if cxx else (steps_device if device_path else steps_host)
    el, (best, poses), syncs_per_step = timed(main_fn, args.steps, args.warmup)
    if (best["template_id"] == D.TOPK_OVERFLOW).any():
        raise RuntimeError("candidate-buffer overflow inside the timed steps")
    if cxx:                                                              # a python-driven device step for the ICP launch's HIP events
        steps_device(1)
        sync_all()
    icp_ms = ev_icp[0].elapsed_time(ev_icp[1]) if device_path or cxx else None
    compare = None
    if (device_path or cxx) and args.compare_host_merge:
        el_h, (best_hm, poses_hm), syncs_h = timed(steps_host, args.steps, 1)
        compare = dict(ms_per_step=round(el_h / args.steps * 1e3, 4), host_syncs_per_step=round(syncs_h, 2),
                       same_result=bool(best_hm.tobytes() == best.tobytes() and np.array_equal(poses_hm.view(np.uint32), poses.view(np.uint32))),
                       note="round-2 path: records to the host, numpy merge, fl_refine_matches, all-reduce from host memory")
    times = None
    roofline = None
    cpu = None
    if world == 1:
        # N = 1: the step is the whole Recognition of the default bench with the hand-over done by select / refine_selected;
        # roofline of its longer kernel like the default line's (SURVEY 8(d) numerators)
        det.match_batch_submit(bptr, dptr, thr)
        det.match_batch_collect(0, 1)                                        # synchronises and reads the stage times
        times = det.stage_times()
        own = np.nonzero(best["template_id"] >= 0)[0]
        jobs = best[own].copy()
        res = det.refine_matches(own.tolist(), jobs, K, params) if len(own) else []
        a = np.frombuffer(res, dtype=api.RESULT_DTYPE) if len(own) else None
        iter_bytes = 0.0 if a is None else float((a["det"]["icp"]["iters"].astype(np.float64) * a["det"]["n_points"] * 72)[a["found"] > 0].sum())
        tpl, _, _ = shard.arrays()
        LM = shard.levels * shard.modalities
        crop_px = float(sum(int(tpl[int(t_) * LM]["width"]) * int(tpl[int(t_) * LM]["height"]) for t_, f_ in zip(jobs["template_id"], a["found"]) if f_)) if a is not None else 0.0
        icp_bytes = iter_bytes + 2 * 14 * crop_px                           # what the fused kernel needs (bench.py: roofline.frac)
        kern = {"k_scan": (times["scan_ms"], times["scan_algorithmic_bytes"]), "k_icp_pipeline": (icp_ms or 0.0, icp_bytes)}
        dom = max(kern, key=lambda q: kern[q][0])
        ach = kern[dom][1] / (kern[dom][0] * 1e-3) / 1e9 if kern[dom][0] > 0 else 0.0
        roofline = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                        traffic=None, launch_ms=round(kern[dom][0], 4), algorithmic_bytes_per_launch=kern[dom][1],
                        numerator="iters*n*72 (SURVEY 8(d)) + 2*14*w*h of the two template-sized crops per refined frame" if dom == "k_icp_pipeline" else "SURVEY 8(d): N*B_tmpl per frame",
                        note="HIP events on the launch stream; PMC traffic is collected for the default line only (profiles/)")
        if cpu_baseline is not None and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, bank, scenes, K)
    verified = None
    mismatches = []
    if getattr(args, "verify_sharded", False) and rank == 0:
        full = api.Detector(ctx, 2, T)
        full.add_class(bank)
        # capacity for every coarse cell of every template at once: the queued entry point used here cannot grow its buffers
        full.finalize(w, h, max_batch=B, max_candidates=max(args.max_candidates, min(1 << 20, n_total * 2048)))
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
    dev_ids = [int(ctx.device)]
    if dist is not None:
        ids = [None] * world
        dist.all_gather_object(ids, int(ctx.device))
        dev_ids = ids
    cap_now = det.grow_candidates(B)                                        # no overflow pending: just reports the capacity
    if mg is not None:
        mg.close()
    det.close()
    return {
        "metric": "frames/sec (640x480 RGB-D x N templates, 20 ICP iters)",
        "value": round(B * args.steps / el, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {w}x{h}, {n_total} templates sharded {args.templates} per GPU over {world} GPUs, "
                               f"{args.levels} pyramid levels T={T}, top-{k} all-gather, winner selected on the device and refined by "
                               f"its owner ({args.icp_iters} ICP iterations forced), ICP mode {args.icp_mode}",
                   "frames_per_step": B, "templates_total": n_total, "templates_per_gpu": args.templates, "levels": args.levels,
                   "parallelism": f"template-sharded x{world}", "candidate_capacity": cap_now,
                   "note": "weak scaling in TEMPLATES: the frames are the same on every rank, the bank grows with the ranks"},
        "collectives": {"backend": (dist.get_backend() if dist is not None else None), "ranks": world, "device_ids": dev_ids,
                        "all_gather_bytes_per_step": world * rec_bytes, "all_reduce_bytes_per_step": B * 17 * 4,
                        "path": "C++ host (libfealess_mg.so: ncclAllGather / ncclAllReduce queued on the context's stream from C++, one host wait per step)" if cxx
                                else "device (select + refine on the GPU, collectives on a second stream, pipelined one step deep)" if device_path
                                else "host merge (records and poses staged through the host)",
                        "host_syncs_per_step": round(syncs_per_step, 2),
                        "host_merge_comparison": compare},
        "icp_ms_last_step": (round(icp_ms, 4) if icp_ms is not None else None),
        "stage_ms_last_step": ({q: round(v, 4) for q, v in times.items() if q.endswith("_ms")} if times else None),
        "detections": f"{found}/{B}", "winner_owner_histogram": owners, "verified_against_single_detector": verified,
        "verify_mismatches": (mismatches[:4] if verified is False else None),
        "roofline": roofline, "cpu_baseline": cpu,
    }
