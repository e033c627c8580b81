#!/usr/bin/env python3
"""bench.py -- frames/s of the FEALESS hot path (LINEMOD detection + ICP refinement) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  A *step* is one pass of the whole
per-frame path -- quantise 2 modalities x L levels, spread/response/linearise, scan N templates,
refine, sort/unique, crop back-projection, ICP, 4x4 pose (CObjRecoLmICP::Recognition,
CadReco/obj_reco_lmicp.cpp:86-204) -- over one batch of synthetic RGB-D frames already resident
in HBM.

Default workload = the shape BASELINE.json's north_star quotes its target on: 640x480, 2000 templates,
2 pyramid levels (T = {5, 8}), 20 ICP iterations (configs[1] with 2000 instead of 360 templates; the
360-template figure is carried in the same line as `c2_360_templates`).  `--config c3` benches
BASELINE configs[2]: 1280x720, T = {5, 8, 4}, 2000 templates, Detector::match only (linemod.cpp:1356-1441).
Ranks shard FRAMES by default (weak scaling, no collective in the data path); `--shard templates` is BASELINE
configs[3]: every rank holds templates/N templates and sees the same frames, the per-rank top-k records are
all-gathered (RCCL), merged as one Detector::match would order them, and the rank that owns the winning template
refines it (ICP); see fealess_amd/bench_sharded.py.

The JSON line carries `roofline` (dominant kernel, algorithmic bytes of SURVEY.md section 8(d)
over the launch duration measured with HIP events on the launch stream) and `cpu_baseline` (the
oracle = CPU restatement of the reference, timed on this box's host cores on a bounded sample).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
L2_PEAK_GBS = 34500.0        # same guide: aggregate L2 bandwidth over the 8 XCDs
FORCE_ALL_ITERS = -3.0e38    # dist_diff_thr that never stops the loop: exactly icp_it_thr iterations
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["c2", "c3"], default="c2",
                    help="c2: 640x480, 2 levels, whole Recognition (default); c3: 1280x720, 3 levels, Detector::match only")
    ap.add_argument("--templates", type=int, default=2000)
    ap.add_argument("--levels", type=int, default=0, help="pyramid levels (default: 2 for c2, 3 for c3)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step per GPU (default 4096 for c2 = four rounds of 4 ICP workgroups on each of the 256 CUs; 256 for c3)")
    ap.add_argument("--icp-iters", type=int, default=20)
    ap.add_argument("--icp-mode", choices=["parity", "fast", "plane"], default="parity")
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--shard", choices=["frames", "templates"], default="frames")
    ap.add_argument("--topk", type=int, default=64, help="records per rank in the template-sharded all-gather")
    ap.add_argument("--match-threshold", type=float, default=75.0,
                    help="--shard templates: Detector::match threshold (the overflow rehearsal of the tests lowers it)")
    ap.add_argument("--max-candidates", type=int, default=4096, help="initial per-frame capacity of the candidate / match buffers")
    ap.add_argument("--sharded-host-merge", action="store_true",
                    help="--shard templates: the round-2 path (records and poses staged through the host, numpy merge) instead of the device one")
    ap.add_argument("--compare-host-merge", action="store_true",
                    help="--shard templates: also time the host-merge path and report it beside the device path's figure")
    ap.add_argument("--mg-host", choices=["python", "cxx"], default="python",
                    help="--shard templates: who drives the collectives -- torch.distributed from Python, or the C++ host "
                         "libfealess_mg.so (include/fealess_mg.h: ncclAllGather / ncclAllReduce from C++; needs one GPU per rank)")
    ap.add_argument("--verify-sharded", action="store_true",
                    help="--shard templates: rank 0 also runs one detector over the whole bank and checks every frame's result bit for bit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra figures of the default line (360 templates, eager front-end, batch sweep, PCIe-inclusive)")
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
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins the process group (gloo), one all-reduce, rank 0 prints the line's "
                         "launch fields -- the CPU test of the --gpus N self-launch")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launch (0 = pick a free one)")
    a = ap.parse_args()
    if a.levels == 0:
        a.levels = 3 if a.config == "c3" else 2
    if a.batch == 0:
        a.batch = 256 if a.config == "c3" else 4096
    return a


def geometry(args):
    if args.config == "c3":
        return 1280, 720, (915.0, 915.0, 640.0, 360.0)
    return 640, 480, (608.0, 608.0, 320.0, 240.0)


def t_pyramid(levels):
    return [5, 8, 4][:levels] if levels == 3 else [5, 8][:levels]


def build_bank(ctx, args, n_templates, w, h, K, spread_trained=False):
    """Scenes + template bank.  Templates = rendered views of the object near each scene pose,
    trained with the product's own addTemplate (fl_extract_template_pyramid; they win the detection and feed ICP with
    real clouds), padded with random templates (the scan's work is data-independent).  Same bank on every rank.
    spread_trained: the trained views are dealt evenly over the bank instead of leading it, so that a template-sharded
    run has winners (and ICP work) on every rank."""
    from fealess_amd import synth
    from fealess_amd.bank import TemplateBank
    levels = args.levels
    fx, fy, cx, cy = K

    def trained_template(Rv, tv, seed):
        d_bg, bgr_v, mask = synth.render(w, h, Rv, tv, seed=seed, noise=False, background=True, fx=fx, fy=fy, cx=cx, cy=cy)
        ex = ctx.extract_template_pyramid(bgr_v, d_bg, (mask * 255).astype(np.uint8), levels)
        if ex is None:
            return None
        d_obj, _, _ = synth.render(w, h, Rv, tv, seed=seed, noise=False, background=False, fx=fx, fy=fy, cx=cx, cy=cy)
        return ex[0], synth.pose13(Rv, tv), (d_obj.astype(np.uint32) * 10).clip(0, 65535).astype(np.uint16)

    rng = np.random.default_rng(1234)
    scenes = []
    trained = []
    for s in range(args.scenes):
        R, t = synth.object_pose(tx=float(rng.uniform(-60, 60)), ty=float(rng.uniform(-40, 40)),
                                 tz=float(rng.uniform(620, 700)), yaw=float(rng.uniform(-0.4, 0.4)),
                                 tilt=float(rng.uniform(0.25, 0.45)), roll=float(rng.uniform(-0.1, 0.2)))
        depth, bgr, _ = synth.render(w, h, R, t, seed=100 + s, fx=fx, fy=fy, cx=cx, cy=cy)
        scenes.append((bgr, depth))
        for v in range(2):
            dR = synth.rot_z(np.deg2rad(rng.uniform(-2, 2))) @ synth.rot_x(np.deg2rad(rng.uniform(-2, 2)))
            tt = t + np.array([rng.uniform(-20, 20), rng.uniform(-15, 15), rng.uniform(-8, 8)])
            out = trained_template(dR @ R, tt, seed=1000 + 10 * s + v)
            if out is not None and len(trained) < n_templates:
                trained.append(out)
    # where the trained views sit in the bank: leading it (default), or every (n / trained)-th pyramid
    nt = len(trained)
    at = {((j * n_templates) // nt if spread_trained else j): j for j in range(nt)}
    bank = TemplateBank("obj", levels, 2)
    for i in range(n_templates):
        if i in at:
            bank.add_pyramid(*trained[at[i]])
        else:                                     # no depth render: they never win against the trained views
            bank.add_pyramid(synth.random_pyramid(rng, levels, 2, w, h), None, None)
    return bank, scenes


def build_clutter(ctx, args, n_templates, w, h, K, n_scenes=6):
    """The cluttered workload: every scene holds THREE instances of the object in front of a textured, non-planar background
    (synth.render_clutter: colour labels on about half of the pixels instead of 1 - 2 %, depth labels that change all over the
    image), and the bank holds four near-by rendered views per instance (1 - 2 degrees / a few mm apart: several of them pass
    threshold 75 on the same instance, so the scan's candidates, the marked tiles and the refinement all grow), padded with
    random templates.  What the reference's per-template loop costs regardless of content (linemod.cpp:1471-1506) and what
    its refinement costs per candidate (:1509-1573)."""
    from fealess_amd import synth
    from fealess_amd.bank import TemplateBank
    levels = args.levels
    fx, fy, cx, cy = K
    rng = np.random.default_rng(4321)
    scenes, trained = [], []
    slots = [(-170.0, -50.0), (10.0, 60.0), (175.0, -35.0)]
    for s in range(n_scenes):
        poses = []
        for (sx, sy) in slots:
            R, t = synth.object_pose(tx=sx + float(rng.uniform(-25, 25)), ty=sy + float(rng.uniform(-25, 25)), tz=float(rng.uniform(620, 720)),
                                     yaw=float(rng.uniform(-0.6, 0.6)), tilt=float(rng.uniform(0.2, 0.5)), roll=float(rng.uniform(-0.15, 0.25)))
            poses.append((R, t))
        depth, bgr, _ = synth.render_clutter(w, h, poses, seed=500 + s, fx=fx, fy=fy, cx=cx, cy=cy)
        scenes.append((bgr, depth))
        for j, (R, t) in enumerate(poses):
            for v in range(4):
                dR = synth.rot_z(np.deg2rad(rng.uniform(-1.5, 1.5))) @ synth.rot_x(np.deg2rad(rng.uniform(-1.5, 1.5)))
                tt = t + np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-5, 5)])
                seed = 5000 + 100 * s + 10 * j + v
                d_bg, bgr_v, mask = synth.render(w, h, dR @ R, tt, seed=seed, noise=False, background=True, fx=fx, fy=fy, cx=cx, cy=cy)
                ex = ctx.extract_template_pyramid(bgr_v, d_bg, (mask * 255).astype(np.uint8), levels)
                if ex is None or len(trained) >= n_templates:
                    continue
                d_obj, _, _ = synth.render(w, h, dR @ R, tt, seed=seed, noise=False, background=False, fx=fx, fy=fy, cx=cx, cy=cy)
                trained.append((ex[0], synth.pose13(dR @ R, tt), (d_obj.astype(np.uint32) * 10).clip(0, 65535).astype(np.uint16)))
    bank = TemplateBank("obj", levels, 2)
    for i in range(n_templates):
        if i < len(trained):
            bank.add_pyramid(*trained[i])
        else:
            bank.add_pyramid(synth.random_pyramid(rng, levels, 2, w, h), None, None)
    return bank, scenes, len(trained)


def build_frames(scenes, B, rank, w, h, same_on_all_ranks=False):
    """Per-rank frames: scene s rolled sideways by a rank/frame dependent even amount (-24 .. +22 px)."""
    bgrs = np.empty((B, h, w, 3), np.uint8)
    depths = np.empty((B, h, w), np.uint16)
    r = 0 if same_on_all_ranks else rank
    for i in range(B):
        b, d = scenes[i % len(scenes)]
        sh = 2 * (((i // len(scenes)) + 5 * r) % 24) - 24
        bgrs[i] = np.roll(b, sh, axis=1)
        depths[i] = np.roll(d, sh, axis=1)
    return bgrs, depths


def cpu_baseline(args, bank, scenes, K):
    """The oracle (CPU restatement of the reference, single thread like the reference) on a bounded
    sample of the same workload: whole Recognition() per frame (c3: Detector::match per frame), same bank,
    same ICP parameters."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    T = t_pyramid(args.levels)

    def one(bgr, depth):
        if args.config == "c3":
            O.match_images(bgr, depth, T, [bank], 75.0)
        else:
            O.recognition(bgr, depth, K, T, bank, 75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, accum64=False, use_kdtree=True)

    n, t0 = 0, time.perf_counter()
    lm_ms = icp_ms = 0.0
    while True:
        one(*scenes[n % len(scenes)])
        if args.config != "c3":
            a, b = O.last_stage_ms()
            lm_ms += a
            icp_ms += b
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 64:
            break
    what = "Detector::match" if args.config == "c3" else f"Recognition, {args.icp_iters} ICP iterations"
    single = dict(value=n / el, unit="frames/s", cores=1, kind="port",
                  sample=f"{n} frames of the same workload ({bank.n_pyramids} templates, {what}), "
                         f"oracle/liboracle.so single-threaded (the reference is single-threaded), {el:.1f} s")
    if args.config != "c3":
        # the reference's own timer points: "Time of linemod" = Detector::match (CadReco/obj_reco_lmicp.cpp:88,124-125),
        # "Time of ICP" = the rest of Recognition() up to the pose (:126,201-202)
        single["linemod_ms"] = round(lm_ms / n, 3)
        single["icp_ms"] = round(icp_ms / n, 3)
        single["stage_note"] = ("per frame, at the reference's timer points (obj_reco_lmicp.cpp:125 'Time of linemod', :202 'Time of ICP'); "
                                "the GPU's per-frame figures for the same two stages are gpu_stage_ms_per_frame")
    # SURVEY 8(d) also asks for the restatement over all host cores: frames are independent, one per thread
    # (ctypes releases the GIL during the call; the oracle keeps no shared mutable state)
    import concurrent.futures as cf
    cores, cores_src = host_cores()
    per_thread = max(1, int(round(n / el * args.cpu_seconds / 2)))       # about cpu_seconds / 2 of work per thread

    def work(k):
        for i in range(per_thread):
            one(*scenes[(k + i) % len(scenes)])
        return per_thread

    t1 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    el2 = time.perf_counter() - t1
    single["all_cores"] = dict(value=done / el2, unit="frames/s", cores=cores,
                               sample=f"{done} frames, one frame per thread at a time on all {cores} host cores this process may use "
                                      f"({cores_src}), {el2:.1f} s")
    single["note"] = ("a scalar port: roughly half of its time is front-end filtering that OpenCV's SIMD kernels do an order of "
                      "magnitude faster, so value / cpu_baseline.value says little about the reference itself")
    return single


def host_cores():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where one is set (a
    one-GPU share of a bigger machine shows all of its cores in the mask but gets only its quota of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    src = f"os.sched_getaffinity: {n}"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    q = max(1, int(round(int(txt[0]) / int(txt[1]))))
                    if q < n:
                        n, src = q, f"cgroup cpu.max {txt[0]}/{txt[1]} of {src}"
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0 and quota // period < n:
                    n, src = max(1, quota // period), f"cgroup cfs quota {quota}/{period} of {src}"
            break
        except Exception:
            continue
    env = os.environ.get("FL_BENCH_CPU_THREADS")
    if env:
        n, src = max(1, int(env)), "FL_BENCH_CPU_THREADS"
    if n > 64:                                   # no quota visible on a big shared host: a one-GPU share is 16 cores (task statement)
        n, src = 16, f"capped: {src}, no cgroup quota visible; one GPU's share of the host is 16 cores"
    return max(1, n), src


def source_digest():
    """Digest of the kernel sources: a PMC file collected for other sources is stale and is not quoted."""
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "fealess_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            hsh.update(name.encode())
            hsh.update(open(os.path.join(d, name), "rb").read())
    return hsh.hexdigest()[:16]


class Runner:
    """One finalized detector + resident device frames of one workload."""

    def __init__(self, ctx, args, bank, bgrs, depths, w, h, K, eager=False, max_candidates=4096, device_frames=None, max_batch=None):
        import torch
        from fealess_amd import api
        from fealess_amd import _lib as L
        self.L, self.torch, self.ctx, self.args = L, torch, ctx, args
        self.w, self.h, self.K, self.bank = w, h, K, bank
        self.det = api.Detector(ctx, 2, t_pyramid(args.levels))
        self.det.add_class(bank)
        ctx.set_option("eager_frontend", 1 if eager else 0)      # sampled by fl_detector_finalize
        self.B = max_batch if max_batch is not None else len(bgrs)
        self.det.finalize(w, h, max_batch=self.B, max_candidates=max_candidates)
        ctx.set_option("eager_frontend", 0)
        if device_frames is not None:                          # frames another Runner already uploaded
            self.d_bgr, self.d_depth = device_frames
        else:
            self.d_bgr = torch.from_numpy(bgrs).cuda()
            self.d_depth = torch.from_numpy(depths.view(np.int16)).cuda()
        torch.cuda.synchronize()
        self.bptr = [self.d_bgr.data_ptr() + i * w * h * 3 for i in range(self.B)]
        self.dptr = [self.d_depth.data_ptr() + i * w * h * 2 for i in range(self.B)]
        mode = {"parity": L.FL_ICP_PARITY, "fast": L.FL_ICP_FAST, "plane": L.FL_ICP_POINT_TO_PLANE}[args.icp_mode]
        self.params = L.RecognitionParams(75.0, args.icp_iters, -1.0, FORCE_ALL_ITERS, mode)

    def step(self, n=None):
        n = self.B if n is None else n
        if self.args.config == "c3":
            self.det.match_batch_submit(self.bptr[:n], self.dptr[:n], 75.0)
        else:
            self.det.recognize_submit_device(self.bptr[:n], self.dptr[:n], self.K, self.params)

    def collect(self, n=None):
        n = self.B if n is None else n
        if self.args.config == "c3":
            got = [self.det.match_batch_collect(i, 64) for i in range(min(n, 8))]
            return got, self.det.stage_times()
        res = self.det.recognize_collect(n)
        return res, self.det.stage_times()

    def timed(self, steps, warmup, n=None, sync=None):
        sync = sync or self.sync
        for _ in range(warmup):
            self.step(n)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(n)
        sync()
        return time.perf_counter() - t0

    def sync(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize()

    def close(self):
        if self.det is not None:
            self.det.close()
            self.det = None
            self.d_bgr = self.d_depth = None
            self.torch.cuda.empty_cache()


def self_launch(args):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE): this parent -- which never imports torch and never
    touches the GPU -- starts the N ranks as a CHILD process (torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1), relays their output and exits with the child's code."""
    import socket
    import subprocess
    port = args.master_port
    if port == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    for line in proc.stdout:                               # rank 0's JSON line (and nothing else) arrives on stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def launch_check(args, world, rank):
    """--launch-check: the distributed plumbing of the line without a GPU (gloo)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_flag": args.gpus,
                          "collectives": {"backend": "gloo" if world > 1 else None, "ranks": world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...), or "
                         f"start `python bench.py --gpus {args.gpus}` plainly and it launches the ranks itself")
    if args.launch_check:
        return launch_check(args, world, rank)
    import torch
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
    ctx = api.Context(local_rank)
    w, h, K = geometry(args)
    T = t_pyramid(args.levels)
    if args.shard == "templates":
        from fealess_amd import bench_sharded
        out = bench_sharded.run(args, ctx, dist, world, rank, w, h, K, build_bank, build_frames, cpu_baseline=cpu_baseline)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        ctx.close()
        return
    bank, scenes = build_bank(ctx, args, args.templates, w, h, K)
    bgrs, depths = build_frames(scenes, args.batch, rank, w, h)
    run = Runner(ctx, args, bank, bgrs, depths, w, h, K, eager=args.eager_frontend)
    B = args.batch

    def sync_all():
        run.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    el = run.timed(args.steps, args.warmup, sync=sync_all)
    res, times = run.collect()
    if dist is not None:
        tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax.item())
    if os.environ.get("FL_ICP_PHASES") and args.config != "c3":   # dev aid (tools/dev): a library built with -DFL_ICP_PHASES returns phase cycles in R
        ph = np.array([[r.det.icp.R[k] for k in range(8)] for r in res]).mean(0)
        print("icp phase Mcycles mean: grid %.2f A1 %.2f A2 %.2f svd %.2f B %.2f | before icp_run %.2f, whole kernel %.2f" %
              (ph[0] / 1e6, ph[2] / 1e6, ph[3] / 1e6, ph[4] / 1e6, ph[5] / 1e6, ph[6] / 1e6, ph[7] / 1e6), file=sys.stderr)
        st = np.array([[r.det.icp.R[8], r.det.icp.T[0], r.det.icp.T[1], r.det.icp.T[2], r.det.icp.dist_mean] for r in res]).mean(0)
        print("icp organised search per frame: steps %.0f, positions per step %.1f (iterations 1-3: %.0f%% of all), fallback steps %.1f%%, "
              "staged points per step %.0f" % (st[0], st[1] / max(st[0], 1), 100 * st[4] / max(st[1], 1), 100 * st[2] / max(st[0], 1),
                                                st[3] / max(st[0], 1)), file=sys.stderr)
        pw = np.array([list(r.pose)[:5] for r in res], np.float64)
        t0 = (pw[:, 0] * 2 ** 24 + pw[:, 1]) / 100.0            # us
        t1 = (pw[:, 2] * 2 ** 24 + pw[:, 3]) / 100.0
        span = t1.max() - t0.min()
        life = t1 - t0
        print("icp workgroup timeline (100 MHz wall clock): launch span %.0f us, workgroup lifetime mean %.0f us (min %.0f, max %.0f), "
              "sum of lifetimes / (span x %d slots) = %.3f; last start %.0f us, first end %.0f us, workgroups ending in the last 10 %% of the span: %d" %
              (span, life.mean(), life.min(), life.max(), min(len(res), 1280), life.sum() / (span * min(len(res), 1280)), t0.max() - t0.min(),
               t1.min() - t0.min(), int((t1 > t0.min() + 0.9 * span).sum())), file=sys.stderr)
        hs = np.array([[0.0] * 5 + list(r.pose)[5:] + list(r.det.R_final) + list(r.det.T_final) for r in res]).sum(0)
        if hs[5:11].sum() > 0:                               # the pipelined search step's own bins (org_search_pipe)
            nst = max(hs[5:8].sum(), 1)
            print("icp pipelined search step: scanned positions per lane and step %.1f, of which inside the lane's own window %.1f; "
                  "batches per row 1 / 2 / 3+: %s %%; staged passes 2 / 3 / 4+: %s %%; tallest lane window 1..6+ rows: %s %%" %
                  (st[1] / max(st[0], 1), st[4] / 64.0 / max(st[0], 1), np.round(100 * hs[5:8] / nst, 1).tolist(),
                   np.round(100 * hs[8:11] / nst, 1).tolist(), np.round(100 * hs[16:22] / nst, 1).tolist()), file=sys.stderr)
            print("icp pipelined search step: widest lane window 1 / 2 / 3 / 4 / 5+ pixels: %s %%" % np.round(100 * hs[11:16] / nst, 1).tolist(), file=sys.stderr)
        stt = hs[11:16].copy() / len(res) / 4e6              # per wave (4 waves per workgroup), M cycles
        hs[11:16] = 0
        a2 = hs[22:25].copy() * 16 / len(res) / 1e6
        hs[22:25] = 0
        print("icp phase A2, M cycles per workgroup: chain wave adding %.2f, chain wave at the tile barrier %.2f, a producer wave at the tile barrier %.2f" % tuple(a2), file=sys.stderr)
        print("icp phase B, M cycles per workgroup: chain wave adding %.2f, chain wave at the tile barrier %.2f, a producer wave at the tile barrier %.2f" % tuple(hs[25:28] * 16 / len(res) / 1e6), file=sys.stderr)
        hs[25:28] = 0
        print("icp search step segments, M cycles per wave: wait for the query %.2f, window + reductions %.2f, staging %.2f, scan %.2f, "
              "epilogue + stores %.2f" % tuple(stt), file=sys.stderr)
        tot = max(hs[:16].sum(), 1)
        print("icp union classes %% of steps, rows W<=13,<=29,<=61,>61 x cols H<=5,<=10,<=20,>20: %s | largest lane window height 1..9+: %s | "
              "width <=4,<=8,<=12: %s" % (np.round(100 * hs[:16].reshape(4, 4) / tot, 1).tolist(), np.round(100 * hs[16:25] / tot, 1).tolist(),
                                          np.round(100 * hs[25:28] / tot, 1).tolist()), file=sys.stderr)
    if os.environ.get("FL_BENCH_NPTS") and args.config != "c3":            # dev aid: how uneven are the jobs of one ICP launch?
        npt = np.array([int(r.det.n_points) for r in res if r.found])
        print("icp n_points: found %d of %d, min / median / mean / max %d / %d / %d / %d, p5 / p95 %d / %d" %
              (len(npt), len(res), npt.min(), np.median(npt), npt.mean(), npt.max(), *np.percentile(npt, [5, 95])), file=sys.stderr)
    frames = B * args.steps * world
    value = frames / el
    if args.config == "c3":
        out = line_c3(args, run, res, times, value, el, world, bank, T)
    else:
        out = line_c2(args, run, res, times, value, el, world, bank, T, bgrs, depths, sync_all)
    dev_ids = [local_rank]
    if dist is not None:
        ids = [None] * world
        dist.all_gather_object(ids, int(torch.cuda.current_device()))
        dev_ids = ids
    out["collectives"] = {"backend": (dist.get_backend() if dist is not None else None), "ranks": world, "device_ids": dev_ids,
                          "data_path": "none (frame-sharded: disjoint frames per rank, whole bank replicated)",
                          "timing": "barrier + all_reduce(MAX) of the ranks' elapsed time" if dist is not None else "single rank"}
    if rank == 0:
        if not args.no_extras and world == 1 and args.config == "c2":
            out.update(extras(args, ctx, run, bank, bgrs, depths, w, h, K))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cb = cpu_baseline(args, bank, scenes, K)
            out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 1)
            if "linemod_ms" in cb and "stage_ms_last_step" in out:
                st = out["stage_ms_last_step"]
                g_lm = (st["total_ms"] - st["icp_ms"]) / B
                g_icp = st["icp_ms"] / B
                cb["gpu_stage_ms_per_frame"] = dict(linemod_ms=round(g_lm, 6), icp_ms=round(g_icp, 6),
                                                    note="device time of the last step / frames: front-end + linear memories + scan + refine + sort, and the ICP launch")
                cb["stage_speedup"] = dict(linemod=round(cb["linemod_ms"] / g_lm, 1), icp=round(cb["icp_ms"] / g_icp, 1))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    run.close()
    ctx.close()


def pmc_traffic(args, bank, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes of tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate passes, same command) -- quoted only when the file was collected for THESE kernel sources and
    this workload; otherwise null (a stale file must never ride along in a fresh record)."""
    try:
        pm = json.load(open(PMC_FILE))
        if pm.get("src_digest") != source_digest() or pm["batch"] != args.batch or pm["templates"] != bank.n_pyramids or \
                args.icp_mode != "parity" or pm.get("config", "c2") != args.config:
            return None, None
        kd = next(v for k, v in pm["kernels"].items() if k.startswith(kernel))
        detail = dict(fetch_size_bytes=kd["FETCH_SIZE"] * 1024, write_size_bytes=kd["WRITE_SIZE"] * 1024, fetch_correction=2.0,
                      source=os.path.relpath(PMC_FILE, ROOT), src_digest=pm["src_digest"],
                      note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB units).  On gfx950 FETCH_SIZE reports "
                           "half of the bytes read (MI355X_MICROARCH.md, HBM section: 128-byte requests tallied at 64); calibrated for "
                           "this kernel's access widths with tools/probes/fetch_calib.hip -- 4, 12 and 16 bytes per lane and 256-byte row "
                           "pieces all read 0.500 of the bytes touched, WRITE_SIZE 1.008 -- so traffic = 2 x FETCH_SIZE + WRITE_SIZE")
        if "SQ_ACTIVE_INST_VALU" in kd and kd.get("GRBM_GUI_ACTIVE"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the waves (one per vector instruction: a wave's issue slot);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs.  A gfx950 SIMD is 32 lanes wide: a wave64 instruction occupies it for 2
            # cycles while one wave issues at most every 4 (MI355X_MICROARCH.md, wave scheduling), so two waves can issue in the
            # same quad-cycle and the pipe is full at 2.0 issue quad-cycles per SIMD quad-cycle
            simd_cycles = kd["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
            issue = 4.0 * kd["SQ_ACTIVE_INST_VALU"] / simd_cycles
            detail["valu_issue_per_simd_quadcycle"] = round(issue, 3)
            detail["valu_pipe_busy"] = round(issue / 2.0, 3)
            detail["valu_note"] = ("valu_issue_per_simd_quadcycle: vector instructions issued per SIMD and quad-cycle (one wave alone can "
                                   "reach 1.0, a SIMD-32 with two or more waves issuing 2.0); valu_pipe_busy = that / 2: the vector "
                                   "ALUs are about half busy and the waves wait half of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES) -- no unit is saturated; "
                                   "DESIGN.md section 4, profiles/README.md")
        return (2.0 * kd["FETCH_SIZE"] + kd["WRITE_SIZE"]) * 1024.0, detail
    except Exception:
        return None, None


def unpruned_scan_ms(run, n=None, steps=3):
    """k_scan's duration with the exact pruning switched off (option scan_prune = 0): the kernel then performs every
    addition SURVEY 8(d)'s N * B_tmpl counts, which is what a bytes-per-second figure must be computed on."""
    run.ctx.set_option("scan_prune", 0)
    try:
        run.timed(steps, 1, n=n)
        _, t = run.collect(n)
    finally:
        run.ctx.set_option("scan_prune", 1)
    return t["scan_ms"]


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


def line_c2(args, run, res, times, value, el, world, bank, T, bgrs, depths, sync_all):
    B = args.batch
    found = sum(int(r.found) for r in res)
    iters = sum(int(r.det.icp.iters) for r in res if r.found)
    npts = sum(int(r.det.n_points) for r in res if r.found)
    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    scan_bytes = times["scan_algorithmic_bytes"]                       # N * B_tmpl per frame, per launch (B frames)
    tpl, _, _ = bank.arrays()
    LM = bank.levels * bank.modalities
    iter_bytes = sum(int(r.det.icp.iters) * int(r.det.n_points) * 72 for r in res if r.found)
    # B_icp = iters*n*(24 corr + 24 transform + 24 dist) + 2*(2+12)*W*H: the reference back-projects both full frames
    icp_bytes_8d = iter_bytes + B * 2 * 14 * run.w * run.h
    # what the fused kernel needs: the same iterations, but only the two template-sized crops are back-projected
    crop_px = sum(int(tpl[int(r.best.template_id) * LM]["width"]) * int(tpl[int(r.best.template_id) * LM]["height"]) for r in res if r.found)
    icp_bytes_need = iter_bytes + 2 * 14 * crop_px
    # `frac` is quoted on the bytes the fused kernel NEEDS (SURVEY 8(d)'s per-iteration figure and the back-projection of the two
    # template-sized crops it performs); 8(d)'s own B_icp also counts the reference's two full-frame back-projections, which the
    # kernel never does -- that larger numerator is carried as `survey_8d`, never as `frac`
    kern = {"k_scan": dict(ms=times["scan_ms"], bytes=scan_bytes), "k_icp_pipeline": dict(ms=times["icp_ms"], bytes=float(icp_bytes_need))}
    dom = max(kern, key=lambda k: kern[k]["ms"])
    ach = gbs(kern[dom]["bytes"], kern[dom]["ms"])
    traffic, traffic_detail = pmc_traffic(args, bank, dom)
    roofline = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic, launch_ms=round(kern[dom]["ms"], 4),
                    algorithmic_bytes_per_launch=kern[dom]["bytes"],
                    numerator=("iters*n*72 per frame (SURVEY 8(d): 24 correspondences + 24 transform + 24 distances per point and iteration) + "
                               "2*14*w*h of the two template-sized crops the fused kernel back-projects") if dom == "k_icp_pipeline"
                    else "SURVEY 8(d): N*B_tmpl per frame",
                    traffic_detail=traffic_detail)
    if traffic:
        moved = gbs(traffic, kern[dom]["ms"])
        roofline["traffic_rate"] = dict(achieved=round(moved, 2), frac=round(moved / HBM_PEAK_GBS, 5), unit="GB/s",
                                        over_algorithmic=round(traffic / kern[dom]["bytes"], 3),
                                        note="the counter traffic over this launch's duration: what the memory side actually moves "
                                             "(Infinity-Cache hits are counted too); achievable HBM is about 6.3 TB/s")
    if dom == "k_icp_pipeline":
        a2 = gbs(icp_bytes_8d, times["icp_ms"])
        roofline["survey_8d"] = dict(achieved=round(a2, 2), frac=round(a2 / HBM_PEAK_GBS, 5), bytes_per_launch=float(icp_bytes_8d),
                                     note="SURVEY 8(d)'s B_icp as written: iters*n*72 + 2*14*W*H per frame, i.e. with the reference's two "
                                          "FULL-FRAME back-projections, which the fused kernel does not perform -- a flattering numerator, "
                                          "kept for comparison with earlier rounds only")
    scan_full_ms = unpruned_scan_ms(run)
    scan_ach = gbs(scan_bytes, scan_full_ms)
    pcie = None
    if args.host_frames_steps > 0 and world == 1 and not args.no_extras:          # like cpu_baseline: N = 1 only
        # informational: the same step with host frames (pinned), i.e. 1.54 MB per frame over PCIe inside the step
        torch = run.torch
        h_bgr = torch.from_numpy(bgrs).pin_memory()
        h_depth = torch.from_numpy(depths.view(np.int16)).pin_memory()
        hb = [h_bgr.data_ptr() + i * run.w * run.h * 3 for i in range(B)]
        hd = [h_depth.data_ptr() + i * run.w * run.h * 2 for i in range(B)]
        run.det.recognize_submit_host(hb, hd, run.K, run.params)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.host_frames_steps):
            run.det.recognize_submit_host(hb, hd, run.K, run.params)
        sync_all()
        el_h = time.perf_counter() - t1
        res_h = run.det.recognize_collect(B)
        pcie = {"value": round(B * args.host_frames_steps * world / el_h, 1), "unit": "frames/s",
                "ms_per_step": round(el_h / args.host_frames_steps * 1e3, 3), "steps": args.host_frames_steps,
                "detections": f"{sum(int(r.found) for r in res_h)}/{B}",
                "note": "frames uploaded from pinned host memory inside every step (2 strided H2D copies on a copy stream, "
                        "double-buffered: batch i+1 uploads while batch i computes); never the headline value"}
    return {
        "metric": "frames/sec (640x480 RGB-D x N templates, 20 ICP iters)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE north_star shape (configs[1] at {bank.n_pyramids} templates): 640x480, {bank.n_pyramids} templates, "
                               f"{args.levels} pyramid levels T={T}, {args.icp_iters} ICP iterations forced (dist_mean_thr=-1, "
                               f"dist_diff_thr=-3e38), ICP mode {args.icp_mode}",
                   "frames_per_step_per_gpu": B, "templates": bank.n_pyramids, "levels": args.levels,
                   "parallelism": f"frame-sharded x{world}",
                   "distinct_frames": f"{min(B, 24 * args.scenes)} ({args.scenes} scenes x 24 sideways shifts of 2 px)",
                   "fine_levels": "eager (whole images, before the scan)" if args.eager_frontend else
                                  "lazy (tiles the scan's candidates touch: data-dependent, see eager_frontend for whole images)"},
        "ms_per_icp_iter": round(times["icp_ms"] / max(1, args.icp_iters), 5),
        "ms_per_icp_iter_per_frame_amortised": round(times["icp_ms"] / max(1, iters), 7),
        "pcie_inclusive": pcie,
        "detections": f"{found}/{B}", "icp_iters_mean": round(iters / max(1, found), 2),
        "icp_points_mean": round(npts / max(1, found), 1),
        "stage_ms_last_step": {k: round(v, 4) for k, v in times.items() if k.endswith("_ms")},
        "roofline": roofline,
        "scan_kernel": {"scan_ms": round(times["scan_ms"], 4), "unpruned_scan_ms": round(scan_full_ms, 4),
                        "achieved_GBs": round(scan_ach, 1), "frac": round(scan_ach / HBM_PEAK_GBS, 4),
                        "l2_frac": round(scan_ach / L2_PEAK_GBS, 4), "l2_peak_GBs": L2_PEAK_GBS,
                        "note": "achieved_GBs / l2_frac: algorithmic bytes (SURVEY 8d N*B_tmpl) over the duration of the UNPRUNED kernel "
                                "(option scan_prune = 0: every addition performed); the linear memories are L2-resident, so the meaningful roof is "
                                "the aggregate L2 bandwidth (l2_frac), not HBM (frac may exceed 1).  scan_ms is the kernel the headline runs: "
                                "it stops a (template, chunk) between the modalities once no position can reach the coarse threshold "
                                "(exact, data-dependent: same match lists)"},
    }


def line_c3(args, run, res, times, value, el, world, bank, T):
    B = args.batch
    scan_bytes = times["scan_algorithmic_bytes"]
    scan_full_ms = unpruned_scan_ms(run)
    scan_ach = gbs(scan_bytes, scan_full_ms)
    stage = {k: round(v, 4) for k, v in times.items() if k.endswith("_ms")}
    dom = max(("frontend_ms", "linmem_ms", "scan_ms", "refine_ms", "lazy_frontend_ms"), key=lambda k: times[k])
    traffic, traffic_detail = pmc_traffic(args, bank, "k_scan")
    roofline = dict(bound="hbm", kernel="k_scan", achieved=round(scan_ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(scan_ach / HBM_PEAK_GBS, 5), traffic=traffic, launch_ms=round(scan_full_ms, 4),
                    pruned_launch_ms=round(times["scan_ms"], 4),
                    algorithmic_bytes_per_launch=scan_bytes, numerator="SURVEY 8(d): N*B_tmpl per frame",
                    l2_frac=round(scan_ach / L2_PEAK_GBS, 5), l2_peak=L2_PEAK_GBS, traffic_detail=traffic_detail,
                    note="launch_ms / achieved / frac: the UNPRUNED kernel (option scan_prune = 0: every addition of N*B_tmpl performed); `value` runs "
                         "the pruning kernel (pruned_launch_ms: a (template, chunk) stops between the modalities once no position can reach "
                         "the threshold -- exact, data-dependent).  The scan's linear memories are L2-resident: judge it against the L2 roof "
                         "(l2_frac); longest stage of the step: " + dom)
    return {
        "metric": "frames/sec (1280x720 RGB-D x N templates, Detector::match only)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: 1280x720, {bank.n_pyramids} templates, {args.levels} pyramid levels T={T}, "
                               "Detector::match only (linemod.cpp:1356-1441; SURVEY M4), threshold 75",
                   "frames_per_step_per_gpu": B, "templates": bank.n_pyramids, "levels": args.levels,
                   "parallelism": f"frame-sharded x{world}"},
        "matches_first_frames": [int(n) for _, n in res],
        "stage_ms_last_step": stage,
        "roofline": roofline,
    }


def extras(args, ctx, run, bank, bgrs, depths, w, h, K):
    """Further figures of the default line (N = 1): batch sweep with the same detector, the 360-template bank of
    BASELINE configs[1] and the data-independent eager front-end -- each a short run of its own."""
    out = {}
    sweep = []
    for b in (1, 8, 64, 256, 1024, 2048, 2560, 4096):
        if b > args.batch:
            continue
        el1 = run.timed(1, 1, n=b)
        steps = int(max(3, min(40, 0.4 / max(el1, 1e-4))))
        el = run.timed(steps, 0, n=b)
        _, t = run.collect(b)
        sweep.append(dict(batch=b, frames_per_s=round(b * steps / el, 1), ms_per_step=round(el / steps * 1e3, 4),
                          icp_ms=round(t["icp_ms"], 4), device_ms_last_step=round(t["total_ms"], 4)))
    out["batch_sweep"] = sweep
    out["batch1_latency_ms"] = sweep[0]["ms_per_step"] if sweep and sweep[0]["batch"] == 1 else None
    # Small batches leave most of the chip idle (the ICP launch of b <= 256 frames occupies b CUs), and the reference's
    # caller hands over one camera frame at a time (test/linemod_recon.cpp:44-111).  Several cameras / streams overlap on one
    # GPU with the entry points that exist for it: K contexts (one HIP stream each) with a detector each, batches submitted
    # round-robin (fl_recognize_submit) and collected K submissions later (fl_recognize_collect).  Same results (same kernels
    # on the same frames); the latency of one batch stays what batch_sweep says.
    from fealess_amd import api as _api
    piped = []
    for b, kpipes in ((1, 8), (8, 8), (64, 4), (256, 2)):
        if b > args.batch:
            continue
        ctxs = [_api.Context(ctx.device) for _ in range(kpipes)]
        runs = [Runner(c, args, bank, None, None, w, h, K, device_frames=(run.d_bgr, run.d_depth), max_batch=b) for c in ctxs]
        for r_ in runs:                                        # warm-up
            r_.step(b)
        for r_ in runs:
            r_.collect(b)
        base = next(e for e in sweep if e["batch"] == b)
        steps = int(max(4 * kpipes, min(400, 0.5 / max(base["ms_per_step"] * 1e-3 / kpipes, 1e-5))))
        t0 = time.perf_counter()
        for i in range(steps):
            r_ = runs[i % kpipes]
            if i >= kpipes:
                r_.collect(b)                                  # the batch this pipeline queued kpipes submissions ago
            r_.step(b)
        last = [r_.collect(b)[0] for r_ in runs]
        el = time.perf_counter() - t0
        found = sum(int(x.found) for x in last[0])
        piped.append(dict(batch=b, pipelines=kpipes, frames_per_s=round(b * steps / el, 1), ms_per_batch_amortised=round(el / steps * 1e3, 4),
                          speedup_vs_one_pipeline=round((b * steps / el) / base["frames_per_s"], 2), detections_last_batch=f"{found}/{b}"))
        for r_ in runs:
            r_.det.close()
            r_.det = None
        for c in ctxs:
            c.close()
    out["batch_sweep_pipelined"] = piped
    out["batch_sweep_pipelined_note"] = ("K contexts x (detector, HIP stream), batches submitted round-robin and collected K submissions later: "
                                         "throughput of several camera streams on one GPU at small batches; per-batch latency is batch_sweep's")
    out["batch_sweep_note"] = ("one Recognition() per camera frame is the reference's call pattern; batches that leave CUs idle run one "
                               "1024-thread ICP workgroup per frame, full batches 256-thread ones (4 or 5 per CU, whichever finishes the batch sooner)")
    if bank.n_pyramids > 360:
        r2 = Runner(ctx, args, bank.subset(0, 360), bgrs, depths, w, h, K)
        el = r2.timed(4, 1)
        _, t = r2.collect()
        out["c2_360_templates"] = dict(value=round(args.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4),
                                       stage_ms={k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
                                       note="BASELINE configs[1] as written: the first 360 pyramids of the same bank")
        r2.close()
    if not args.eager_frontend:
        r3 = Runner(ctx, args, bank, bgrs, depths, w, h, K, eager=True)
        el = r3.timed(4, 1)
        _, t = r3.collect()
        out["eager_frontend"] = dict(value=round(args.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4),
                                     stage_ms={k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
                                     note="finer pyramid levels quantised and spread in full before the scan (the reference's order): the "
                                          "data-independent figure; same results")
        r3.close()
    # the whole step with the scan's exact pruning off: with eager_frontend the data-independent figures
    ctx.set_option("scan_prune", 0)
    try:
        el = run.timed(4, 1)
        _, t = run.collect()
    finally:
        ctx.set_option("scan_prune", 1)
    out["scan_unpruned"] = dict(value=round(args.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4),
                                scan_ms=round(t["scan_ms"], 4),
                                note="option scan_prune = 0: every template's every feature added at every position (the reference's work); "
                                     "same results")
    if not args.eager_frontend:
        # both data-dependent savings off in ONE run: whole-image fine levels AND every addition of the scan -- the figure that
        # does not depend on what the frames show
        r5 = Runner(ctx, args, bank, None, None, w, h, K, eager=True, device_frames=(run.d_bgr, run.d_depth), max_batch=args.batch)
        ctx.set_option("scan_prune", 0)
        try:
            el = r5.timed(4, 1)
            _, t = r5.collect()
        finally:
            ctx.set_option("scan_prune", 1)
        out["eager_unpruned"] = dict(value=round(args.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4),
                                     stage_ms={k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
                                     note="eager front-end AND unpruned scan together: the DATA-INDEPENDENT figure of the whole step (the "
                                          "reference's work on every frame, whatever it shows); same results")
        r5.det.close()
        r5.det = None
    if args.icp_mode == "parity":
        # FL_ICP_FAST (parallel sums instead of the reference's float32 chains): within 1e-4 of the EXACT sums, not of the reference's
        # float32 result (tests/test_gpu_icp.py::test_icp_fast_mode_close_to_fp64_yardstick, ::test_icp_fast_mode_recognition_vs_the_f32_oracle)
        af = argparse.Namespace(**vars(args))
        af.icp_mode = "fast"
        r4 = Runner(ctx, af, bank, bgrs, depths, w, h, K)
        el = r4.timed(4, 1)
        _, t = r4.collect()
        el1 = r4.timed(20, 2, n=1)
        _, t1 = r4.collect(1)
        out["icp_fast"] = dict(value=round(args.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4),
                               icp_ms=round(t["icp_ms"], 4), batch1_latency_ms=round(el1 / 20 * 1e3, 4), batch1_icp_ms=round(t1["icp_ms"], 4),
                               note="--icp-mode fast: float32 per-thread partial sums + fp64 tree instead of the reference-order float32 chains; "
                                    "final pose within 1e-4 of the exact (fp64) sums; up to 1.1e-4 (R) / 1.1e-4 (T relative to the object's "
                                    "distance) from the reference's float32 result on 15 k-point clouds, which is that result's own "
                                    "summation noise (the tests assert <= 2.5e-4 and no farther than the float32 result is from the exact "
                                    "sums); NOT the mode that meets the 1e-4 bar against the reference -- FL_ICP_PARITY does, with 0; never "
                                    "the headline value")
        r4.close()
    # configs[4]'s per-GPU shape on this GPU: 64 frames x 2000 templates as eight batches of 8 (what each of 8 GPUs would run)
    if args.batch >= 64:
        el8 = run.timed(8 * 4, 8, n=8)
        _, t8 = run.collect(8)
        p8 = next((e for e in out.get("batch_sweep_pipelined", []) if e["batch"] == 8), None)
        out["c5_64x2000_batch8"] = dict(frames_per_s=round(8 * 32 / el8, 1), ms_per_64_frames=round(el8 / 4 * 1e3, 4), ms_per_batch_of_8=round(el8 / 32 * 1e3, 4),
                                        icp_ms_per_batch=round(t8["icp_ms"], 4), pipelined_frames_per_s=(p8["frames_per_s"] if p8 else None),
                                        workload=f"BASELINE configs[4] per-GPU shape: 8 frames x {bank.n_pyramids} templates per call, eight calls "
                                                 "back to back = the node's 64-frame batch on one GPU; pipelined_frames_per_s = the same "
                                                 "calls over 8 contexts (batch_sweep_pipelined); parity of this shape: "
                                                 "tests/test_gpu_configs.py::test_c5_64_frames_x_2000_templates_in_batches_of_8_equal_one_batch_and_the_oracle")
    run.close()                                           # the default workload's 26 GB of workspaces make room for the other configs
    out.update(clutter_config(args, ctx, w, h, K))
    out.update(c4_config(args, ctx, w, h, K))
    out.update(other_configs(args, ctx))
    return out


def clutter_config(args, ctx, w, h, K):
    """The cluttered workload beside the headline (same shape: 640x480, 2000 templates, 2 levels, 20 forced ICP iterations)."""
    a = argparse.Namespace(**vars(args))
    a.batch = min(args.batch, 2048)
    bank, scenes, n_trained = build_clutter(ctx, a, a.templates, w, h, K)
    bgrs, depths = build_frames(scenes, a.batch, 0, w, h)
    r = Runner(ctx, a, bank, bgrs, depths, w, h, K, max_candidates=65536)
    el = r.timed(4, 1)
    res, t = r.collect()
    cnt = np.array([r.det.frame_counters(i) for i in range(0, a.batch, max(1, a.batch // 64))], np.float64)
    tiles_total = ((w + 59) // 60) * ((h + 59) // 60)
    out = dict(value=round(a.batch * 4 / el, 1), unit="frames/s", ms_per_step=round(el / 4 * 1e3, 4), frames_per_step=a.batch,
               stage_ms={k: round(v, 4) for k, v in t.items() if k.endswith("_ms")},
               detections=f"{sum(int(x.found) for x in res)}/{a.batch}", overflowed_frames=int(sum(int(x.status) != 0 for x in res)),
               coarse_candidates_per_frame=round(float(cnt[:, 0].mean()), 1), matches_per_frame=round(float(cnt[:, 1].mean()), 1),
               marked_level0_tile_fraction=round(float(cnt[:, 3].mean()) / tiles_total, 3),
               icp_points_mean=round(float(np.mean([int(x.det.n_points) for x in res if x.found] or [0])), 1),
               trained_views=n_trained,
               workload=f"clutter: 640x480, {bank.n_pyramids} templates ({n_trained} rendered views: 4 near-by views of each of 3 object instances "
                        f"in {len(scenes)} scenes, the rest random), textured non-planar background, 3 objects per frame, {a.icp_iters} ICP "
                        "iterations forced on the best match (fealess_amd/synth.py::render_clutter, bench.py::build_clutter)")
    r.ctx.set_option("scan_prune", 0)
    try:
        el2 = r.timed(3, 1)
        _, t2 = r.collect()
    finally:
        r.ctx.set_option("scan_prune", 1)
    out["scan_unpruned_ms"] = round(t2["scan_ms"], 4)
    out["ms_per_step_scan_unpruned"] = round(el2 / 3 * 1e3, 4)
    r.close()
    return {"clutter": out}


def c4_config(args, ctx, w, h, K):
    """BASELINE configs[3]'s bank on ONE GPU: a single detector over 16000 templates (what the eight 2000-template ranks must add
    up to; the sharded path itself is `--shard templates`, its parity tests/test_gpu_configs.py)."""
    a = argparse.Namespace(**vars(args))
    a.templates, a.batch = 16000, min(args.batch, 1024)
    bank, scenes = build_bank(ctx, a, a.templates, w, h, K, spread_trained=True)
    bgrs, depths = build_frames(scenes, a.batch, 0, w, h)
    r = Runner(ctx, a, bank, bgrs, depths, w, h, K)
    el = r.timed(3, 1)
    res, t = r.collect()
    full = unpruned_scan_ms(r, steps=2)
    out = dict(value=round(a.batch * 3 / el, 1), unit="frames/s", ms_per_step=round(el / 3 * 1e3, 4), frames_per_step=a.batch,
               stage_ms={k: round(v, 4) for k, v in t.items() if k.endswith("_ms")}, scan_ms=round(t["scan_ms"], 4), unpruned_scan_ms=round(full, 4),
               scan_l2_frac=round(gbs(t["scan_algorithmic_bytes"], full) / L2_PEAK_GBS, 4),
               detections=f"{sum(int(x.found) for x in res)}/{a.batch}",
               winner_template_ids=sorted({int(x.best.template_id) for x in res if x.found})[:12],
               workload=f"BASELINE configs[3]'s bank on one GPU: 640x480, {bank.n_pyramids} templates in ONE detector (trained views dealt evenly over the "
                        f"bank, global ids up to {bank.n_pyramids - 1}), 2 levels, {a.icp_iters} ICP iterations forced")
    r.close()
    return {"c4_16000_templates_1gpu": out}


def other_configs(args, ctx):
    """BASELINE configs[0] and configs[2] on this GPU, each a short run of its own, so that one driver run records every
    single-GPU configuration (configs[1] is `c2_360_templates`, the headline is configs[1]'s shape at 2000 templates)."""
    out = {}
    # configs[2]: 1280x720, 2000 templates, 3 levels T = {5, 8, 4}, Detector::match only (= `bench.py --config c3`)
    a3 = argparse.Namespace(**vars(args))
    a3.config, a3.levels, a3.batch, a3.templates, a3.icp_mode = "c3", 3, 256, 2000, "parity"
    w3, h3, K3 = geometry(a3)
    bank3, scenes3 = build_bank(ctx, a3, a3.templates, w3, h3, K3)
    b3, d3 = build_frames(scenes3, a3.batch, 0, w3, h3)
    r3 = Runner(ctx, a3, bank3, b3, d3, w3, h3, K3)
    steps = 8
    el = r3.timed(steps, 2)
    res3, t3 = r3.collect()
    line = line_c3(a3, r3, res3, t3, a3.batch * steps / el, el, 1, bank3, t_pyramid(3))
    out["c3_1280x720"] = dict(value=line["value"], unit="frames/s", ms_per_step=round(el / steps * 1e3, 4), frames_per_step=a3.batch,
                              stage_ms=line["stage_ms_last_step"], scan_l2_frac=line["roofline"]["l2_frac"],
                              scan_achieved_GBs=line["roofline"]["achieved"], unpruned_scan_ms=line["roofline"]["launch_ms"],
                              matches_first_frames=line["matches_first_frames"],
                              workload=line["config"]["workload"])
    r3.close()
    del r3, b3, d3
    # configs[0]: one 640x480 frame, 16 templates, 1 pyramid level (T = {5}: the scan runs at level 0), match + ICP
    a1 = argparse.Namespace(**vars(args))
    a1.config, a1.levels, a1.batch, a1.templates, a1.icp_mode = "c2", 1, 256, 16, "parity"
    w1, h1, K1 = geometry(a1)
    bank1, scenes1 = build_bank(ctx, a1, a1.templates, w1, h1, K1)
    b1, d1 = build_frames(scenes1, a1.batch, 0, w1, h1)
    r1 = Runner(ctx, a1, bank1, b1, d1, w1, h1, K1)
    el1 = r1.timed(20, 2, n=1)
    res1, t1 = r1.collect(1)
    elb = r1.timed(4, 1)
    resb, tb = r1.collect()
    out["c1_16_templates_1_level"] = dict(
        batch1_latency_ms=round(el1 / 20 * 1e3, 4), batch1_stage_ms={k: round(v, 4) for k, v in t1.items() if k.endswith("_ms")},
        value=round(a1.batch * 4 / elb, 1), unit="frames/s", frames_per_step=a1.batch, ms_per_step=round(elb / 4 * 1e3, 4),
        stage_ms={k: round(v, 4) for k, v in tb.items() if k.endswith("_ms")},
        detections=f"{sum(int(r.found) for r in resb)}/{a1.batch}",
        workload=f"BASELINE configs[0] on the GPU: 640x480, {bank1.n_pyramids} templates, 1 pyramid level T=[5], Recognition with "
                 f"{a1.icp_iters} ICP iterations forced; one frame per call (batch1_*) and {a1.batch} frames per step")
    r1.close()
    return out


if __name__ == "__main__":
    main()
