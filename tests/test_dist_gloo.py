"""CPU, world_size 2, gloo: the N > 1 path's host logic -- document partitioning,
the all-gather of (rows, thresholds) with ragged sizes, column order of the fold --
with the oracle standing in for the per-rank GPU run and for the fold, checked against
a direct run on the union."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import pyoracle as O
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_haps, length = 6, 6000
    groups = mdist.partition_docs(n_haps, world)
    docs = synth.pangenome_subset(n_haps, length, 0.01, 5, groups[rank])
    r = O.run(docs, merge=True)
    L, off, st = r.mum_rows()
    th = torch.from_numpy(r.thresh()[: length + 1].astype(np.int16))
    parts = mdist.all_gather_partitions((L, off, st, th), dist, torch.device("cpu"))
    if rank != 0:
        assert all(p[0] is None for p in parts) and len(parts) == world
    # the tensor-resident variant (what the GPU ranks use) must hand over the very same partitions
    tparts = mdist.all_gather_partitions_device(
        (torch.from_numpy(L.astype(np.int32)), torch.from_numpy(off), torch.from_numpy(st), th), dist)
    ref = mdist.all_gather_partitions((L, off, st, th), dist, torch.device("cpu"), host_on_rank=None)
    assert len(tparts) == world
    for a, b in zip(tparts, ref):
        assert np.array_equal(a[0].numpy().view(np.uint32), b[0]) and np.array_equal(a[1].numpy(), b[1])
        assert np.array_equal(a[2].numpy(), b[2]) and torch.equal(a[3], b[3])
        assert a[1].is_contiguous() and a[2].is_contiguous()
    if rank == 0:
        host_parts = [(p[0], p[1], p[2], p[3].numpy().view(np.uint16)) for p in parts]
        ml, mo, ms, mth = O.anchor_merge(host_parts)
        q.put((groups, ml, mo, ms, mth))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_exchange_and_fold():
    import torch.multiprocessing as mp
    import pyoracle as O
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth
    from mumsfile import format_mums
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    groups, ml, mo, ms, mth = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert groups == [[0, 1, 2, 3], [0, 4, 5]]
    order = mdist.merged_column_order(groups)
    assert order == [0, 1, 2, 3, 4, 5]
    all_docs = synth.pangenome(6, 6000, 0.01, 5)
    direct = O.run([all_docs[i] for i in order], merge=True)
    anchor = all_docs[0][0]
    idx = sorted(range(len(ml)), key=lambda i: anchor[mo[i, 0]: mo[i, 0] + ml[i]])
    assert format_mums(ml[idx], mo[idx], ms[idx]) == direct.text()
    assert np.array_equal(mth, direct.thresh()[: 6001])


def test_partition_docs_shapes():
    from mumemto_amd import dist as mdist
    assert mdist.partition_docs(94, 8) == [[0] + list(range(1 + 12 * i, 13 + 12 * i)) for i in range(5)] + \
        [[0] + list(range(61 + 11 * i, 72 + 11 * i)) for i in range(3)]
    g = mdist.partition_docs(16, 1)
    assert g == [list(range(16))]
    for n, w in [(5, 2), (9, 4), (31, 8)]:
        gs = mdist.partition_docs(n, w)
        assert sorted(d for g in gs for d in g[1:]) == list(range(1, n)) and all(g[0] == 0 for g in gs)


def _gather_worker(rank, world, port, q):
    sys.path[:0] = [ROOT]
    import torch
    import torch.distributed as dist
    from mumemto_amd import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = [b"first rank\n" * 3, b"", b"third\tand last\n"][rank]
    parts = mdist.gather_bytes(local, dist, torch.device("cpu"))
    q.put((rank, parts))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_bytes_keeps_rank_order_with_ragged_and_empty_parts():
    """The only collective of the range-sharded multi-GPU path (mdist.run_sharded): every rank's output bytes, in rank
    order, on every rank."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(3):
        assert got[r] == [b"first rank\n" * 3, b"", b"third\tand last\n"]


@pytest.mark.parametrize("seed,world", [(5, 2), (6, 3), (7, 5), (8, 8)])
def test_fold_over_anchor_coordinate_ranges_equals_the_pairwise_fold(seed, world):
    """SURVEY 8(e), the reduce-scatter shape: the anchor is cut into `world` slices, every slice is folded on its own
    (rows reaching into it + a left margin of (partitions - 1) row lengths) -- the concatenation is the pairwise fold of
    the whole anchor (src/merge_candidates.cpp:106-157), rows and merged thresholds alike."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import pyoracle as O
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth
    docs = synth.pangenome(9, 12000, 0.01, seed=seed, indel_rate=0.001, inversion=(3, 2000, 4000))
    groups = [[0, 1, 2], [0, 3, 4, 5], [0, 6], [0, 7, 8]]
    parts = []
    for g in groups:
        r = O.run([docs[i] for i in g], merge=True)
        l, o, s = r.mum_rows()
        parts.append((l, o, s, r.thresh()[: 12001]))
    whole = O.anchor_merge(parts)
    split = mdist.fold_by_ranges(O.anchor_merge, parts, world)
    assert len(whole[0]) > 20
    for a, b in zip(whole, split):
        assert np.array_equal(np.asarray(a), np.asarray(b).reshape(np.asarray(a).shape))


def _range_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import pyoracle as O
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_haps, length = 10, 9000
    groups = mdist.partition_docs(n_haps, world)
    docs = synth.pangenome(n_haps, length, 0.01, 9)
    r = O.run([docs[i] for i in groups[rank]], merge=True)
    L, off, st = r.mum_rows()
    merged = mdist.merge_by_ranges(O.anchor_merge, (L, off, st, r.thresh()[: length + 1]), dist, torch.device("cpu"), length + 1)
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_fold_over_coordinate_ranges_with_threshold_all_to_all():
    """mdist.merge_by_ranges on 3 ranks (gloo): rows all-gathered, thresholds exchanged slice-wise (all-to-all), every rank
    folds its slice of the anchor, pieces gathered -- equal to the pairwise fold of the three partitions on one rank."""
    import torch.multiprocessing as mp
    import pyoracle as O
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    docs = synth.pangenome(10, 9000, 0.01, 9)
    parts = []
    for g in mdist.partition_docs(10, world):
        r = O.run([docs[i] for i in g], merge=True)
        l, o, s_ = r.mum_rows()
        parts.append((l, o, s_, r.thresh()[: 9001]))
    ml, mo, ms, _ = O.anchor_merge(parts)
    assert len(ml) > 10
    for rank in range(world):
        gl, go, gs = got[rank]
        assert np.array_equal(gl, ml) and np.array_equal(go, mo) and np.array_equal(gs, ms)


def test_native_slice_arithmetic_equals_the_python_fold_and_tiles_the_anchor():
    """mmt_fold_slice_bounds (merge.cpp fold_slice_bounds / fold_margin: what every rank of dist_merge_ranges and every
    slice of mmt_anchor_merge_by_ranges computes) on the host, no device: the slices tile [0, L) in rank order, each reads
    from lo - ((k - 1) x longest + 1) (clamped at 0), and they are the slices of mdist.fold_slices / fold_margin -- also for
    more ranks than anchor positions, eight ranks of a whole-genome anchor and a margin longer than a slice."""
    import ctypes as C
    from mumemto_amd import binding
    from mumemto_amd import dist as mdist
    L = binding.load_library()
    for thresh_len, world, k, longest in ((12001, 3, 3, 800), (5, 8, 8, 3), (3_050_000_001, 8, 8, 65535),
                                          (64_000_001, 4, 4, 40_000_000), (1, 1, 2, 0), (1000, 7, 2, 20)):
        margin = mdist.fold_margin([(np.array([longest], np.uint32),)] * k)
        assert margin == (k - 1) * longest + 1
        want = mdist.fold_slices(thresh_len, world)
        at = 0
        for r in range(world):
            b = (C.c_uint64 * 3)()
            assert L.mmt_fold_slice_bounds(thresh_len, world, r, k, longest, b) == 0
            assert (b[0], b[1]) == want[r] and b[0] == at and b[2] == max(0, b[0] - margin)
            at = b[1]
        assert at == thresh_len
    b = (C.c_uint64 * 3)()
    assert L.mmt_fold_slice_bounds(100, 4, 4, 2, 10, b) != 0          # rank out of range
