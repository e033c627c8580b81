"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The reference is single-threaded and has no communication layer; the path shards in two
independent ways:

* frame-sharded  -- every rank holds the whole template bank and its own frames; no data-path
  collective at all (this is what bench.py measures, "scaling": "weak").
* template-sharded -- every rank holds a contiguous slice of the bank, all ranks see the same
  frame; the one real exchange step is an all-gather of each rank's fixed-size top-k match
  records (k * 20 bytes, latency-bound), after which every rank merges them exactly as one
  Detector::match over the whole bank would order them (std::sort + std::unique,
  linemod.cpp:1437-1439).  Contiguous slices keep template ids ordered across ranks, which is what
  makes "local sort/unique, then merge" equal to the global result.  Recognition() then refines
  matches[0] (obj_reco_lmicp.cpp:111-197): the rank whose slice holds that template does it (the
  depth renders are sharded with the templates) and the poses are combined over the ranks: every
  frame has exactly one owner, the others contribute zero rows, and the rows are summed as INT32
  bit patterns (x + 0 = x exactly; a float sum would turn an owner's -0.0 into +0.0)
  (`template_sharded_recognize`).

A frame whose candidate buffers overflowed on some rank is exported as record 0 =
{template_id = TOPK_OVERFLOW}; the flag travels with the all-gather, so every rank sees it and all
ranks take the same branch: grow the buffers and run the step again, or report the frame as failed.
"""
import numpy as np

from .bank import MATCH_DTYPE

TOPK_OVERFLOW = -2            # include/fealess_hip.h FL_TOPK_OVERFLOW


def shard_range(n, world, rank):
    """Contiguous, balanced slice [first, first+count) of n templates for `rank`."""
    base, rem = divmod(n, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def shard_bank(bank, world, rank):
    first, count = shard_range(bank.n_pyramids, world, rank)
    return bank.subset(first, count), first


def pad_topk(records, k, template_id_base=0):
    """First k records with global template ids, padded with template_id = -1 (host-side twin of
    fl_export_topk, used by the CPU tests)."""
    out = np.zeros(k, MATCH_DTYPE)
    out["template_id"] = -1
    out["class_idx"] = -1
    n = min(k, len(records))
    out[:n] = records[:n]
    out["template_id"][:n] += template_id_base
    return out


def allgather_records(local, dist, device=None):
    """All-gather of fixed-size record buffers.  `local` is a numpy MATCH_DTYPE array (gloo / CPU
    tests) or a torch uint8 device tensor filled by fl_export_topk (nccl / RCCL).  Returns a numpy
    MATCH_DTYPE array of world * k records."""
    import torch
    world = dist.get_world_size()
    if isinstance(local, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).copy())
        if device is not None:
            t = t.to(device)
    else:
        t = local
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().view(MATCH_DTYPE)


def template_sharded_match(det, ctx, bgr, depth, threshold, k, template_id_base, dist):
    """One frame matched against a bank sharded over the ranks: local match on this rank's
    detector, fl_export_topk into a device buffer, RCCL all-gather, exact merge on every rank."""
    import torch
    from .api import merge_topk
    det.match(bgr, depth, threshold, cap=1)                  # leaves the sorted matches in HBM
    buf = torch.empty(k * MATCH_DTYPE.itemsize, dtype=torch.uint8, device=f"cuda:{ctx.device}")
    det.export_topk(0, k, template_id_base, buf.data_ptr())
    ctx.synchronize()
    gathered = allgather_records(buf, dist)
    return merge_topk(gathered, k)


def owner_of(template_id, n_templates, world):
    """Rank whose contiguous slice (shard_range) holds global template id `template_id`."""
    for r in range(world):
        first, count = shard_range(n_templates, world, r)
        if first <= template_id < first + count:
            return r
    raise ValueError(f"template {template_id} outside 0..{n_templates}")


def owners_of(template_ids, n_templates, world):
    """Vectorised owner_of: int array of ranks (-1 where the id is negative)."""
    firsts = np.array([shard_range(n_templates, world, r)[0] for r in range(world)], np.int64)
    t = np.asarray(template_ids, np.int64)                         # negative ids (none / TOPK_OVERFLOW) have no owner
    return np.where(t >= 0, np.searchsorted(firsts, t, side="right") - 1, -1).astype(np.int32)


def best_of_ranks(gathered):
    """matches[0] of one global std::sort over all ranks' lists, per frame, without merging the lists: every rank's list is
    already sorted, so the global first element is the first of the ranks' first elements in the same order
    (similarity descending, then template id, class, y, x ascending -- fl_merge_topk's comparator).
    gathered: MATCH_DTYPE [world, n_frames, k]; returns MATCH_DTYPE [n_frames]: template_id = -1 where no rank matched,
    TOPK_OVERFLOW where some rank's candidate buffers overflowed (host twin of fl_select_best_batch)."""
    heads = np.ascontiguousarray(gathered[:, :, 0])                 # [world, n_frames]
    world, n_frames = heads.shape
    valid = heads["template_id"] >= 0
    over = (heads["template_id"] == TOPK_OVERFLOW).any(axis=0)
    # np.lexsort: last key is the primary one
    sim = np.where(valid, heads["similarity"], -np.inf)
    order = np.lexsort((heads["x"].T, heads["y"].T, heads["class_idx"].T, np.where(valid, heads["template_id"], np.iinfo(np.int32).max).T,
                        -sim.T), axis=1)                            # [n_frames, world]: ranks in match order
    win = order[:, 0]
    best = heads[win, np.arange(n_frames)].copy()
    none = ~valid.any(axis=0) | over
    best["x"][none] = 0
    best["y"][none] = 0
    best["similarity"][none] = 0
    best["class_idx"][none] = -1
    best["template_id"][none] = -1
    best["template_id"][over] = TOPK_OVERFLOW
    return best


def template_sharded_recognize(n_frames, k, n_templates, world, rank, local_topk, allgather, refine, allreduce_sum, full_lists=False,
                               grow=None, max_attempts=6):
    """Recognition() of n_frames frames against a bank sharded over `world` ranks -- the host logic, with the device work
    behind callables so that the CPU tests (gloo + oracle) and the GPU path (RCCL + HIP detector) run the same code:

      local_topk()            -> this rank's records of all frames, MATCH_DTYPE [n_frames, k] with GLOBAL template ids,
                                 padded with template_id = -1 (numpy), or a uint8 device tensor of the same bytes; a frame
                                 whose candidate buffers overflowed carries TOPK_OVERFLOW in record 0 (fl_export_topk_batch)
      allgather(local)        -> numpy MATCH_DTYPE [world, n_frames, k]
      refine(frames, matches) -> [n_jobs, 17] float32: found flag + row-major 4x4 pose, for the frames whose winning template
                                 this rank owns (`matches` carry class-local ids of this rank's shard)
      allreduce_sum(array)    -> the elementwise sum over ranks of an INT32 numpy array (every rank gets it)
      grow()                  -> (optional) grow this rank's candidate buffers to what its last batch needs
                                 (fl_detector_grow_candidates); called on EVERY rank when any rank overflowed -- the flag is
                                 in the gathered records, so all ranks agree -- after which the step runs again.  May raise
                                 (hard cap, no memory): the failure is all-reduced, every rank stops retrying, and the
                                 frames concerned are reported as TOPK_OVERFLOW

    Recognition() only ever uses matches[0] (obj_reco_lmicp.cpp:111), which is the best of the ranks' best records
    (best_of_ranks); full_lists=True also merges the whole lists as one Detector::match would order them
    (fl_merge_topk_batch) and returns their lengths (capped at k), otherwise n_matches is None.

    Returns (first match per frame, MATCH_DTYPE [n_frames], global ids, template_id = -1 where nothing matched and
             TOPK_OVERFLOW where a rank's list overflowed and could not be grown (that frame: found = 0);
             n_matches per frame or None; poses float32 [n_frames, 17])."""
    first, _count = shard_range(n_templates, world, rank)
    for attempt in range(max_attempts + 1):
        gathered = np.ascontiguousarray(allgather(local_topk())).reshape(world, n_frames, k)
        best = best_of_ranks(gathered)
        if not (best["template_id"] == TOPK_OVERFLOW).any() or grow is None or attempt == max_attempts:
            break
        # every rank: same records, same decision to grow.  The OUTCOME of growing is made collective too: a rank whose
        # buffers cannot grow (hard cap, out of memory: fl_detector_grow_candidates -> FL_ERR_OVERFLOW) must not leave the
        # others waiting in the next all-gather, so the failure count is summed over the ranks and everyone stops together;
        # the frames concerned stay TOPK_OVERFLOW (found = 0).
        failed = 0
        try:
            grow()
        except Exception:                                      # noqa: BLE001 -- whatever it was, the ranks must agree on it
            failed = 1
        if int(np.asarray(allreduce_sum(np.array([failed], np.int32))).reshape(-1)[0]) != 0:
            break
    n_out = None
    ok = best["template_id"] != TOPK_OVERFLOW
    if full_lists:
        from .api import merge_topk_batch
        merged, n_out = merge_topk_batch(gathered.reshape(-1), world, n_frames, k, k)
        has = (n_out > 0) & ok
        assert np.array_equal(merged[has, 0], best[has]) and (best["template_id"][~has] < 0).all()
    # one global std::sort + unique, then matches[0] (linemod.cpp:1437-1439, obj_reco_lmicp.cpp:111): its owner refines it
    mine = np.nonzero(owners_of(best["template_id"], n_templates, world) == rank)[0]
    poses = np.zeros((n_frames, 17), np.float32)
    if len(mine):
        jobs = best[mine].copy()
        jobs["template_id"] -= first
        poses[mine] = refine(mine.tolist(), jobs)
    # exactly one rank owns a frame: the sum of the int32 bit patterns IS the owner's row, bit for bit
    total = np.ascontiguousarray(allreduce_sum(np.ascontiguousarray(poses).view(np.int32)), np.int32)
    return best, n_out, total.view(np.float32).reshape(n_frames, 17)
