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


def all_gather_partitions(local, dist, device, host_on_rank=0):
    """local = (length u32[n], offsets i64[n,nd], strands u8[n,nd], thresh) where thresh is a
    1-D int16 torch tensor on `device` (the anchor thresholds, L_0+1 entries).
    Four collectives per step (sizes, offsets+lengths, strands, thresholds), each an RCCL
    all-gather of padded, exactly typed buffers.  Only rank `host_on_rank` (None = every rank)
    copies the gathered rows back to host numpy arrays -- the other ranks get thresholds only.
    Returns the list of every rank's tuple (thresh as torch tensor on `device`)."""
    import torch
    length, off, st, thresh = local
    world = dist.get_world_size()
    rank = dist.get_rank()
    n, nd = off.shape if off.ndim == 2 else (0, 0)
    meta = torch.tensor([n, nd], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    max_n = max(max(m[0] for m in metas), 1)
    max_nd = max(max(m[1] for m in metas), 1)
    # i64 table [length | offsets...] and u8 table [strands...], padded to the largest partition
    tab = np.zeros((max_n, 1 + max_nd), np.int64)
    stt = np.zeros((max_n, max_nd), np.uint8)
    if n:
        tab[:n, 0] = length
        tab[:n, 1:1 + nd] = off
        stt[:n, :nd] = st
    t = torch.from_numpy(tab).to(device)
    u = torch.from_numpy(stt).to(device)
    tables = [torch.empty_like(t) for _ in range(world)]
    stables = [torch.empty_like(u) for _ in range(world)]
    dist.all_gather(tables, t)
    dist.all_gather(stables, u)
    # thresholds travel as raw bytes: neither RCCL nor gloo has a 16-bit integer type
    tbytes = thresh.contiguous().view(torch.uint8)
    gathered = [torch.empty_like(tbytes) for _ in range(world)]
    dist.all_gather(gathered, tbytes)
    threshes = [g.view(torch.int16) for g in gathered]
    want_host = host_on_rank is None or rank == host_on_rank
    parts = []
    for r in range(world):
        rn, rnd = metas[r]
        if want_host:
            tb = tables[r][:rn].cpu().numpy()
            sb = stables[r][:rn].cpu().numpy()
            parts.append((tb[:, 0].astype(np.uint32), np.ascontiguousarray(tb[:, 1:1 + rnd]),
                          np.ascontiguousarray(sb[:, :rnd]), threshes[r]))
        else:
            parts.append((None, None, None, threshes[r]))
    return parts


def all_gather_partitions_device(local, dist):
    """The same exchange without leaving the device the tensors live on.
    local = (length int32[n], offsets int64[n,nd], strands uint8[n,nd], thresh int16[L_0+1]) as torch
    tensors (HBM under RCCL; CPU tensors under gloo).  Returns, for every rank, a tuple
    (length, offsets, strands, thresh) of contiguous tensors on that same device."""
    import torch
    length, off, st, thresh = local
    world = dist.get_world_size()
    device = off.device
    n, nd = (int(off.shape[0]), int(off.shape[1])) if off.ndim == 2 else (0, 0)
    meta = torch.tensor([n, nd], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    max_n = max(max(m[0] for m in metas), 1)
    max_nd = max(max(m[1] for m in metas), 1)
    # padded to the largest partition; one typed collective per table
    tl = torch.zeros(max_n, dtype=torch.int32, device=device)
    to = torch.zeros((max_n, max_nd), dtype=torch.int64, device=device)
    ts = torch.zeros((max_n, max_nd), dtype=torch.uint8, device=device)
    if n:
        tl[:n] = length.view(torch.int32)
        to[:n, :nd] = off
        ts[:n, :nd] = st
    gl = [torch.empty_like(tl) for _ in range(world)]
    go = [torch.empty_like(to) for _ in range(world)]
    gs = [torch.empty_like(ts) for _ in range(world)]
    dist.all_gather(gl, tl)
    dist.all_gather(go, to)
    dist.all_gather(gs, ts)
    tbytes = thresh.contiguous().view(torch.uint8)
    gt = [torch.empty_like(tbytes) for _ in range(world)]
    dist.all_gather(gt, tbytes)
    parts = []
    for r in range(world):
        rn, rnd = metas[r]
        parts.append((gl[r][:rn].contiguous(), go[r][:rn, :rnd].contiguous(), gs[r][:rn, :rnd].contiguous(),
                      gt[r].view(torch.int16)))
    return parts


def device_partitions(parts):
    """binding.DevicePartition objects over the tensors all_gather_partitions_device returned."""
    from .binding import DevicePartition
    return [DevicePartition(p[1].shape[0], p[1].shape[1], p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(),
                            p[3].data_ptr(), p[3].numel(), keepalive=p) for p in parts]


class DevicePointerView:
    """Exposes a raw HBM pointer through __cuda_array_interface__ so that torch can
    wrap the engine's buffers (thresholds, row tables) without a copy."""

    def __init__(self, ptr, shape, typestr="<i2"):
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def engine_rows_as_tensors(eng, device):
    """(length int32[n], offsets int64[n,nd], strands uint8[n,nd]) torch views of the engine's last MUM rows
    in HBM -- no copy; valid until the engine's next run."""
    import torch
    n, nd, lp, op, sp = eng.rows_mum_device()
    if n == 0:
        return (torch.empty(0, dtype=torch.int32, device=device), torch.empty((0, nd), dtype=torch.int64, device=device),
                torch.empty((0, nd), dtype=torch.uint8, device=device))
    return (torch.as_tensor(DevicePointerView(lp, (n,), "<i4"), device=device),
            torch.as_tensor(DevicePointerView(op, (n, nd), "<i8"), device=device),
            torch.as_tensor(DevicePointerView(sp, (n, nd), "|u1"), device=device))


def gather_bytes(local, dist, device):
    """Every rank's byte string on every rank, in rank order (one all-gather of the sizes, one of the padded bytes)."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = max(max(sizes), 1)
    buf = torch.zeros(width, dtype=torch.uint8, device=device)
    if len(local):
        buf[: len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [out[r][: sizes[r]].cpu().numpy().tobytes() for r in range(world)]


def run_sharded(eng, dist, device, **params):
    """Partial multi-MUMs / multi-MEMs (and any other mode) on several GPUs without a partition merge -- which the
    reference refuses for these modes (include/pfp_mum.hpp:178-183): every rank holds the whole collection, builds the
    tables of the parse and produces, scans and drops only its share of the SA / LCP / BWT stream (Engine.set_scan_shard:
    no column is exchanged and none is stored); rows
    come out in order of their closing position, so the ranks' .mums / .mems bytes concatenated in rank order are the
    bytes of a single-GPU run.  The only collective is the gather of those bytes.  The input must have been set on
    every rank (set_docs / set_input_device).  Returns the whole output on every rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    eng.set_scan_shard(rank, world)
    try:
        eng.run(**params)
        local = eng.output_text()
    finally:
        eng.set_scan_shard(0, 1)
    return b"".join(gather_bytes(local, dist, device))


def run_sort_sharded(eng, dist, device, **params):
    """The same through the bucket-wise producer (suffixes bucketed by their leading characters sort
    independently, the buckets in order are the suffix array, and every reportable interval lies inside one bucket): a
    rank sorts, scans and drops only the bins of its share -- its device memory holds the text, the tables of the parse
    and one batch, whatever the size of the collection."""
    eng.set_producer("guided")
    try:
        return run_sharded(eng, dist, device, **params)
    finally:
        eng.set_producer("auto")



# ---- fold over anchor coordinate ranges (the reduce-scatter shape) ------------------------------------
# The pairwise fold (src/merge_candidates.cpp:106-157) is local in the anchor coordinate: what it decides at position i
# depends on the thresholds at i and on the row of each side that covers i.  A slice [lo, hi) of the anchor can therefore
# be folded on its own from the thresholds of the slice and the rows that reach into it -- plus a margin to the left:
# a row of an intermediate result that covers lo may start up to one row length before lo, and was itself decided from
# rows up to one more length further left, once per fold step.  Margin = (partitions - 1) x the longest row.


def fold_slices(n_positions, world):
    """`world` near-equal slices [lo, hi) of the anchor positions 0 .. n_positions - 1."""
    cuts = [n_positions * r // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def fold_margin(parts):
    """Positions to the left of a slice that its fold has to see: (partitions - 1) x the longest row, + 1."""
    longest = max([int(np.asarray(p[0]).max()) if len(p[0]) else 0 for p in parts] + [0])
    return (len(parts) - 1) * longest + 1


def fold_slice(fold, parts, lo, hi, thresh_base=None):
    """The rows of fold(parts) that start in [lo, hi) and its thresholds [lo, hi), computed from that slice alone.
    parts: (length, offsets[n, nd], strands[n, nd], thresh) per partition; fold: parts -> merged tuple.  thresh is the
    whole array (L0 + 1 entries), or -- with thresh_base = max(0, lo - fold_margin(parts)) -- just thresh[base:hi]."""
    base = max(0, lo - fold_margin(parts))
    assert thresh_base is None or thresh_base == base
    sliced = []
    for length, off, st, th in parts:
        off = np.asarray(off).reshape(len(length), -1)
        st = np.asarray(st).reshape(len(length), -1)
        a0 = off[:, 0] if len(length) else np.zeros(0, np.int64)
        keep = (a0 >= base) & (a0 < hi)
        o = off[keep].copy()
        o[:, 0] -= base
        th = np.asarray(th)
        sliced.append((np.asarray(length)[keep], o, st[keep], th[base:hi] if thresh_base is None else th))
    ml, mo, ms, mth = fold(sliced)
    mo = np.asarray(mo).reshape(len(ml), -1).copy()
    ms = np.asarray(ms).reshape(len(ml), -1)
    if len(ml):
        mo[:, 0] += base
        mine = (mo[:, 0] >= lo) & (mo[:, 0] < hi)
        ml, mo, ms = np.asarray(ml)[mine], mo[mine], ms[mine]
    return ml, mo, ms, np.asarray(mth)[lo - base:hi - base]


def fold_by_ranges(fold, parts, world):
    """fold(parts), computed as `world` independent slices of the anchor (what `world` ranks do in parallel)."""
    n_positions = len(parts[0][3])
    pieces = [fold_slice(fold, parts, lo, hi) for lo, hi in fold_slices(n_positions, world)]
    nd = sum(np.asarray(p[1]).reshape(len(p[0]), -1).shape[1] - 1 for p in parts) + 1
    ml = np.concatenate([p[0] for p in pieces]) if pieces else np.zeros(0, np.uint32)
    mo = np.concatenate([p[1].reshape(-1, nd) for p in pieces])
    ms = np.concatenate([p[2].reshape(-1, nd) for p in pieces])
    return ml, mo, ms, np.concatenate([p[3] for p in pieces])


def merge_by_ranges(fold, local, dist, device, n_positions):
    """The multi-GPU fold in the reduce-scatter shape: rows (small) are all-gathered; of the thresholds (2 bytes per
    anchor position and rank) every rank receives only its slice of the anchor, with the margin of fold_slice, from every
    other rank (one all-to-all); every rank folds its slice with `fold` (its own GPU); the pieces -- rows that start in
    the slice -- are gathered in rank order.  local = (length u32[n], offsets i64[n, nd], strands u8[n, nd], thresh
    u16[n_positions]) as numpy arrays.  Returns (length, offsets, strands) of the whole fold on every rank, in anchor order
    (Engine.rows_in_direct_order re-sorts them)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    length, off, st, thresh = local
    rows = [None] * world
    dist.all_gather_object(rows, (np.asarray(length), np.asarray(off).reshape(len(length), -1),
                                  np.asarray(st).reshape(len(length), -1)))
    margin = fold_margin([(r[0],) for r in rows])
    slices = fold_slices(n_positions, world)
    bases = [max(0, lo - margin) for lo, _ in slices]
    th = torch.from_numpy(np.ascontiguousarray(thresh, np.uint16).view(np.int16)).to(device)
    send = [th[bases[r]:slices[r][1]].contiguous() for r in range(world)]
    recv = [torch.empty(slices[rank][1] - bases[rank], dtype=torch.int16, device=device) for _ in range(world)]
    try:
        dist.all_to_all(recv, send)            # RCCL
    except RuntimeError:                       # gloo has no all-to-all: the same exchange as point-to-point messages
        ops = []
        for r in range(world):
            if r != rank:
                ops.append(dist.P2POp(dist.isend, send[r], r))
                ops.append(dist.P2POp(dist.irecv, recv[r], r))
        recv[rank].copy_(send[rank])
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
    parts = [(rows[g][0], rows[g][1], rows[g][2], recv[g].cpu().numpy().view(np.uint16)) for g in range(world)]
    lo, hi = slices[rank]
    piece = fold_slice(fold, parts, lo, hi, thresh_base=bases[rank])
    pieces = [None] * world
    dist.all_gather_object(pieces, piece[:3])
    nd = sum(r[1].shape[1] - 1 for r in rows) + 1
    return (np.concatenate([p[0] for p in pieces]), np.concatenate([p[1].reshape(-1, nd) for p in pieces]),
            np.concatenate([p[2].reshape(-1, nd) for p in pieces]))

