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
  makes "local sort/unique, then merge" equal to the global result.
"""
import numpy as np

from .bank import MATCH_DTYPE


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
