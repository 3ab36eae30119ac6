"""Multi-GPU driver: the reference's own partition -> independent runs -> anchor
merge workflow (README "merge", src/merge_candidates.cpp, mumemto/merge_mums.py
:58-93,171-192) with the files between partitions replaced by one RCCL
all-gather of (MUM rows, thresholds) over xGMI.

Rank r runs the single-GPU hot path on {anchor} + group_r with merge metadata
on; every rank then holds every partition's rows and thresholds, rank 0 folds
them on its GPU (bit-identical to anchor_merge) and re-sorts the rows into the
order of a direct run.  No collective touches the SA/LCP data path.
"""
import numpy as np


def partition_docs(n_docs, world):
    """Doc 0 is the anchor, replicated; docs 1..n-1 are dealt to `world` groups
    in contiguous, near-equal blocks.  Returns list of index lists."""
    rest = list(range(1, n_docs))
    base, extra = divmod(len(rest), world)
    groups, k = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        groups.append([0] + rest[k:k + size])
        k += size
    return groups


def merged_column_order(groups):
    """Column order of the fold: partition 0's docs, then every other partition's
    docs without its anchor (merge_candidates.cpp:142-151)."""
    return groups[0] + [d for g in groups[1:] for d in g[1:]]


def all_gather_partitions(local, dist, device):
    """local = (length u32[n], offsets i64[n,nd], strands u8[n,nd], thresh) where thresh is a
    1-D int16 torch tensor on `device` (the anchor thresholds, L_0+1 entries).
    Returns the list of every rank's tuple (thresh as torch tensor on `device`).
    One all_gather for the sizes, one for the padded rows, one for the thresholds."""
    import torch
    length, off, st, thresh = local
    world = dist.get_world_size()
    n, nd = off.shape if off.ndim == 2 else (0, 0)
    meta = torch.tensor([n, nd], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    max_n = max(m[0] for m in metas)
    max_nd = max(m[1] for m in metas)
    # one padded i64 table per rank: [length | offsets... | strands...]
    table = np.zeros((max(max_n, 1), 1 + 2 * max(max_nd, 1)), np.int64)
    if n:
        table[:n, 0] = length
        table[:n, 1:1 + nd] = off
        table[:n, 1 + max_nd:1 + max_nd + nd] = st
    t = torch.from_numpy(table).to(device)
    tables = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(tables, t)
    # thresholds travel as raw bytes: neither RCCL nor gloo has a 16-bit integer type
    tbytes = thresh.contiguous().view(torch.uint8)
    gathered = [torch.zeros_like(tbytes) for _ in range(world)]
    dist.all_gather(gathered, tbytes)
    threshes = [g.view(torch.int16) for g in gathered]
    parts = []
    for r in range(world):
        rn, rnd = metas[r]
        tb = tables[r].cpu().numpy()
        parts.append((tb[:rn, 0].astype(np.uint32), tb[:rn, 1:1 + rnd].copy(),
                      tb[:rn, 1 + max_nd:1 + max_nd + rnd].astype(np.uint8), threshes[r]))
    return parts


class DevicePointerView:
    """Exposes a raw HBM pointer through __cuda_array_interface__ so that torch can
    wrap the engine's threshold buffer without a copy."""

    def __init__(self, ptr, n, typestr="<i2"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
