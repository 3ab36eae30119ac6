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
