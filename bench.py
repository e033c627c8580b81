#!/usr/bin/env python3
"""bench.py -- frames/s of the FEALESS hot path (LINEMOD detection + ICP refinement) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  A *step* is one pass of the whole
per-frame path -- quantise 2 modalities x L levels, spread/response/linearise, scan N templates,
refine, sort/unique, crop back-projection, ICP, 4x4 pose (CObjRecoLmICP::Recognition,
CadReco/obj_reco_lmicp.cpp:86-204) -- over one batch of synthetic RGB-D frames already resident
in HBM.  Default workload = BASELINE.json configs[1]: 640x480, 360 templates, 2 pyramid levels
(T = {5, 8}), 20 ICP iterations.  Ranks shard FRAMES (weak scaling): every rank holds the whole
template bank and its own batch; there is no collective in the data path.

The JSON line carries `roofline` (dominant kernel, algorithmic bytes of SURVEY.md section 8(d)
over the launch duration measured with HIP events on the launch stream) and `cpu_baseline` (the
oracle = CPU restatement of the reference, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
FORCE_ALL_ITERS = -3.0e38    # dist_diff_thr that never stops the loop: exactly icp_it_thr iterations


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--templates", type=int, default=360)
    ap.add_argument("--levels", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1280, help="frames per step per GPU (1280 = 5 ICP workgroups on each of the 256 CUs)")
    ap.add_argument("--icp-iters", type=int, default=20)
    ap.add_argument("--icp-mode", choices=["parity", "fast", "plane"], default="parity")
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-frames-steps", type=int, default=6,
                    help="extra untimed-for-value steps with the frames in pinned HOST memory (upload inside the step): the "
                         "PCIe-inclusive rate reported as pcie_inclusive (0 = skip)")
    ap.add_argument("--eager-frontend", action="store_true",
                    help="quantise and spread the finer pyramid levels in full before the scan (the reference's order) instead of "
                         "only in the tiles the scan's candidates touch; same results")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (implies a non-RCCL backend)")
    return ap.parse_args()


def build_workload(ctx, args, rank):
    """Synthetic frames + template bank.  Templates = rendered views of the
    object near each scene pose, trained with the product's own addTemplate (fl_extract_template_pyramid; they
    win the detection and feed ICP with real clouds), padded with random templates (the scan's work is
    data-independent).  The oracle is never touched outside the cpu_baseline leg."""
    from fealess_amd import synth
    from fealess_amd.bank import TemplateBank
    levels = args.levels
    w, h = 640, 480

    def trained_template(Rv, tv, seed):
        """One training view through the product's own Detector::addTemplate (fl_extract_template_pyramid): the
        rendered object with its mask -> template pyramid, 13-float pose, depth render in 0.1 mm."""
        d_bg, bgr_v, mask = synth.render(w, h, Rv, tv, seed=seed, noise=False, background=True)
        ex = ctx.extract_template_pyramid(bgr_v, d_bg, (mask * 255).astype(np.uint8), levels)
        if ex is None:
            return None
        d_obj, _, _ = synth.render(w, h, Rv, tv, seed=seed, noise=False, background=False)
        return ex[0], synth.pose13(Rv, tv), (d_obj.astype(np.uint32) * 10).clip(0, 65535).astype(np.uint16)

    rng = np.random.default_rng(1234)        # same bank on every rank
    bank = TemplateBank("obj", levels, 2)
    scenes = []
    for s in range(args.scenes):
        R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)),
                                 tz=float(rng.uniform(620, 700)), yaw=float(rng.uniform(-0.4, 0.4)),
                                 tilt=float(rng.uniform(0.25, 0.45)), roll=float(rng.uniform(-0.1, 0.2)))
        depth, bgr, _ = synth.render(w, h, R, t, seed=100 + s)
        scenes.append((bgr, depth))
        for v in range(2):
            dR = synth.rot_z(np.deg2rad(rng.uniform(-2, 2))) @ synth.rot_x(np.deg2rad(rng.uniform(-2, 2)))
            tt = t + np.array([rng.uniform(-20, 20), rng.uniform(-15, 15), rng.uniform(-8, 8)])
            out = trained_template(dR @ R, tt, seed=1000 + 10 * s + v)
            if out is not None and bank.n_pyramids < args.templates:
                bank.add_pyramid(*out)
    zero = np.zeros((h, w), np.uint16)
    while bank.n_pyramids < args.templates:
        bank.add_pyramid(synth.random_pyramid(rng, levels, 2, w, h), None, zero)
    # per-rank frames: scene s rolled sideways by a rank/frame dependent even amount
    B = args.batch
    bgrs = np.empty((B, h, w, 3), np.uint8)
    depths = np.empty((B, h, w), np.uint16)
    for i in range(B):
        b, d = scenes[i % len(scenes)]
        sh = 2 * (((i // len(scenes)) + 3 * rank) % 12) - 12
        bgrs[i] = np.roll(b, sh, axis=1)
        depths[i] = np.roll(d, sh, axis=1)
    return bank, bgrs, depths, scenes


def cpu_baseline(args, bank, scenes):
    """The oracle (CPU restatement of the reference, single thread like the reference) on a bounded
    sample of the same workload: whole Recognition() per frame, same bank, same ICP parameters."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    T = [5, 8, 4][:args.levels] if args.levels == 3 else [5, 8][:args.levels]
    K = (608.0, 608.0, 320.0, 240.0)
    n, t0 = 0, time.perf_counter()
    while True:
        bgr, depth = scenes[n % len(scenes)]
        O.recognition(bgr, depth, K, T, bank, 75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, accum64=False, use_kdtree=True)
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 64:
            break
    single = dict(value=n / el, unit="frames/s", cores=1, kind="port",
                  sample=f"{n} frames of the same workload ({bank.n_pyramids} templates, {args.icp_iters} ICP iterations), "
                         f"oracle/liboracle.so single-threaded (the reference is single-threaded), {el:.1f} s")
    # SURVEY 8(d) also asks for the restatement over all host cores: frames are independent, one per thread
    # (ctypes releases the GIL during the call; the oracle keeps no shared mutable state)
    import concurrent.futures as cf
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))                                        # the CPU share of a one-GPU box
    per_thread = max(1, int(round(n / el * args.cpu_seconds / 2)))       # about cpu_seconds / 2 of work per thread

    def work(k):
        for i in range(per_thread):
            bgr, depth = scenes[(k + i) % len(scenes)]
            O.recognition(bgr, depth, K, T, bank, 75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, accum64=False, use_kdtree=True)
        return per_thread

    t1 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    el2 = time.perf_counter() - t1
    single["all_cores"] = dict(value=done / el2, unit="frames/s", cores=cores,
                               sample=f"{done} frames, one frame per thread at a time, {el2:.1f} s")
    return single


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            local_rank = 0
            dist.init_process_group("gloo" if args.dist_backend == "nccl" else args.dist_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    from fealess_amd import api
    from fealess_amd import _lib as L
    ctx = api.Context(local_rank)
    T = [5, 8, 4] if args.levels == 3 else [5, 8][:args.levels]
    bank, bgrs, depths, scenes = build_workload(ctx, args, rank)
    det = api.Detector(ctx, 2, T)
    det.add_class(bank)
    if args.eager_frontend:
        os.environ["FL_EAGER_FRONTEND"] = "1"      # read by fl_detector_finalize
    det.finalize(640, 480, max_batch=args.batch, max_candidates=4096)
    B = args.batch
    d_bgr = torch.from_numpy(bgrs).cuda()
    d_depth = torch.from_numpy(depths.view(np.int16)).cuda()
    torch.cuda.synchronize()
    bptr = [d_bgr.data_ptr() + i * 640 * 480 * 3 for i in range(B)]
    dptr = [d_depth.data_ptr() + i * 640 * 480 * 2 for i in range(B)]
    K = (608.0, 608.0, 320.0, 240.0)
    mode = {"parity": L.FL_ICP_PARITY, "fast": L.FL_ICP_FAST, "plane": L.FL_ICP_POINT_TO_PLANE}[args.icp_mode]
    params = L.RecognitionParams(75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, mode)

    def step():
        det.recognize_submit_device(bptr, dptr, K, params)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    el = time.perf_counter() - t0
    res = det.recognize_collect(B)
    times = det.stage_times()            # HIP events on the launch stream, last step
    if dist is not None:
        tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    if os.environ.get("FL_ICP_PHASES"):   # dev aid (tools/dev): a library built with -DFL_ICP_PHASES returns phase cycles in R
        ph = np.array([[r.det.icp.R[k] for k in range(8)] for r in res]).mean(0)
        print("icp phase Mcycles mean: grid %.2f A1check %.2f A1search %.2f A2 %.2f svd %.2f B %.2f | before icp_run %.2f, whole kernel %.2f" %
              (ph[0] / 1e6, ph[1] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6, ph[5] / 1e6, ph[6] / 1e6, ph[7] / 1e6),
              file=sys.stderr)
        st = np.array([[r.det.icp.R[8], r.det.icp.T[0], r.det.icp.T[1], r.det.icp.T[2], r.det.icp.dist_mean] for r in res]).mean(0)
        print("icp organised search per frame: steps %.0f, positions per step %.1f (iterations 1-3: %.0f%% of all), fallback steps %.1f%%, "
              "staged points per step %.0f" % (st[0], st[1] / max(st[0], 1), 100 * st[4] / max(st[1], 1), 100 * st[2] / max(st[0], 1),
                                                st[3] / max(st[0], 1)), file=sys.stderr)
    pcie = None
    if args.host_frames_steps > 0 and world == 1:          # like cpu_baseline: N = 1 only
        # informational: the same step with host frames (pinned), i.e. 1.54 MB per frame over PCIe inside the step
        h_bgr = torch.from_numpy(bgrs).pin_memory()
        h_depth = torch.from_numpy(depths.view(np.int16)).pin_memory()
        hb = [h_bgr.data_ptr() + i * 640 * 480 * 3 for i in range(B)]
        hd = [h_depth.data_ptr() + i * 640 * 480 * 2 for i in range(B)]
        det.recognize_submit_host(hb, hd, K, params)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.host_frames_steps):
            det.recognize_submit_host(hb, hd, K, params)
        sync_all()
        el_h = time.perf_counter() - t1
        res_h = det.recognize_collect(B)
        pcie = {"value": round(B * args.host_frames_steps * world / el_h, 1), "unit": "frames/s",
                "ms_per_step": round(el_h / args.host_frames_steps * 1e3, 3), "steps": args.host_frames_steps,
                "detections": f"{sum(int(r.found) for r in res_h)}/{B}",
                "note": "frames uploaded from pinned host memory inside every step (2 strided H2D copies on a copy stream, "
                        "double-buffered: batch i+1 uploads while batch i computes); never the headline value"}
    found = sum(int(r.found) for r in res)
    iters = sum(int(r.det.icp.iters) for r in res if r.found)
    npts = sum(int(r.det.n_points) for r in res if r.found)
    frames = B * args.steps * world
    value = frames / el

    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    scan_bytes = times["scan_algorithmic_bytes"]                       # N * B_tmpl per frame, per launch (B frames)
    # B_icp = iters*n*(24 corr + 24 transform + 24 dist) + 2*(2+12)*W*H back-projection, per frame
    icp_bytes = sum(int(r.det.icp.iters) * int(r.det.n_points) * 72 for r in res if r.found) + B * 2 * 14 * 640 * 480
    kern = {
        "k_scan": dict(ms=times["scan_ms"], bytes=scan_bytes),
        "k_icp_pipeline": dict(ms=times["icp_ms"], bytes=float(icp_bytes)),
    }
    dom = max(kern, key=lambda k: kern[k]["ms"])
    ach = kern[dom]["bytes"] / (kern[dom]["ms"] * 1e-3) / 1e9 if kern[dom]["ms"] > 0 else 0.0
    # HBM traffic of that kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE, separate passes, same command); only quoted when the workload matches
    traffic = None
    traffic_detail = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc_b1280.json")))
        if pm["batch"] == B and pm["templates"] == bank.n_pyramids and args.icp_mode == "parity":
            kd = pm["kernels"][dom if dom != "k_icp_pipeline" else "k_icp_pipeline<0>"]
            traffic = (kd["FETCH_SIZE"] + kd["WRITE_SIZE"]) * 1024.0          # bytes per launch, raw counters
            traffic_detail = dict(fetch_bytes=kd["FETCH_SIZE"] * 1024, write_bytes=kd["WRITE_SIZE"] * 1024,
                                  source="profiles/r01_final_pmc_b1280.json",
                                  note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; gfx950 FETCH_SIZE "
                                       "under-counts wide coalesced reads by up to 2x (this kernel's reads are mostly "
                                       "4-16 B gathers: uncalibrated), so true HBM reads lie between 1x and 2x fetch_bytes")
    except Exception:
        traffic = None
    roofline = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic,
                    launch_ms=round(kern[dom]["ms"], 4), algorithmic_bytes_per_launch=kern[dom]["bytes"],
                    traffic_detail=traffic_detail)
    scan_ach = scan_bytes / (times["scan_ms"] * 1e-3) / 1e9 if times["scan_ms"] > 0 else 0.0

    out = None
    if rank == 0:
        out = {
            "metric": "frames/sec (640x480 RGB-D x N templates, 20 ICP iters)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 640x480, {bank.n_pyramids} templates, {args.levels} pyramid "
                                   f"levels T={T}, {args.icp_iters} ICP iterations forced (dist_mean_thr=-1, "
                                   f"dist_diff_thr=-3e38), ICP mode {args.icp_mode}",
                       "frames_per_step_per_gpu": B, "templates": bank.n_pyramids, "levels": args.levels,
                       "parallelism": f"frame-sharded x{world}",
                       "fine_levels": "eager (whole images, before the scan)" if args.eager_frontend else
                                      "lazy (tiles the scan's candidates touch; --eager-frontend for whole images)"},
            "ms_per_icp_iter": round(times["icp_ms"] / max(1, args.icp_iters), 5),
            "ms_per_icp_iter_per_frame_amortised": round(times["icp_ms"] / max(1, iters), 7),
            "pcie_inclusive": pcie,
            "detections": f"{found}/{B}", "icp_iters_mean": round(iters / max(1, found), 2),
            "icp_points_mean": round(npts / max(1, found), 1),
            "stage_ms_last_step": {k: round(v, 4) for k, v in times.items() if k.endswith("_ms")},
            "roofline": roofline,
            "scan_kernel": {"achieved_GBs": round(scan_ach, 1), "frac": round(scan_ach / HBM_PEAK_GBS, 4),
                            "note": "algorithmic bytes; linear memories are L2-resident so this may exceed HBM peak"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, bank, scenes)
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    det.close()
    ctx.close()


if __name__ == "__main__":
    main()
