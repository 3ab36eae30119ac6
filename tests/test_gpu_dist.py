"""GPU: the multi-GPU step's device plumbing on one MI355X -- thresholds wrapped
zero-copy as a torch tensor, RCCL all-gather (world_size 1), fold on the GPU with
device-resident thresholds, re-sort to direct-run order."""
import os
import socket

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth

pytestmark = pytest.mark.gpu


def test_rccl_exchange_and_device_fold():
    import torch
    import torch.distributed as dist
    import mumemto_amd
    from mumemto_amd import dist as mdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        device = torch.device("cuda", 0)
        docs = synth.pangenome(7, 40000, 0.01, seed=17)
        groups = [[0, 1, 2, 3], [0, 4, 5, 6]]
        L0 = len(docs[0][0])
        eng = mumemto_amd.Engine(0, torch.cuda.current_stream(device).cuda_stream)
        parts = []
        for g in reversed(groups):      # the last run must hold the anchor ranks; any partition does
            eng.set_docs([docs[i] for i in g])
            eng.run(merge_metadata=True)
            length, off, st = eng.rows_mum()
            th = torch.as_tensor(mdist.DevicePointerView(eng.thresh_device_ptr(), L0 + 1), device=device)
            torch.cuda.synchronize()
            gathered = mdist.all_gather_partitions((length, off, st, th), dist, device)
            assert len(gathered) == 1
            p = gathered[0]
            assert np.array_equal(p[0], length) and np.array_equal(p[1], off) and np.array_equal(p[2], st)
            parts.append((p[0], p[1], p[2], p[3].clone()))
        parts.reverse()
        merged = eng.anchor_merge([(p[0], p[1], p[2], (p[3].data_ptr(), L0 + 1)) for p in parts],
                                  sort_like_direct=True)
        order = mdist.merged_column_order(groups)
        direct = O.run([docs[i] for i in order], merge=True)
        assert merged["text"] == direct.text()
        assert np.array_equal(merged["thresh"], direct.thresh()[: L0 + 1])
        eng.close()
    finally:
        dist.destroy_process_group()


def test_device_resident_exchange_and_fold():
    """The bench's N > 1 step: rows and thresholds never leave HBM between the per-partition run, the
    RCCL all-gather and the fold; only the merged .mums bytes come back."""
    import torch
    import torch.distributed as dist
    import mumemto_amd
    from mumemto_amd import dist as mdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        device = torch.device("cuda", 0)
        docs = synth.pangenome(10, 30000, 0.01, seed=23)
        groups = [[0, 1, 2, 3], [0, 4, 5], [0, 6, 7, 8, 9]]
        L0 = len(docs[0][0])
        eng = mumemto_amd.Engine(0, torch.cuda.current_stream(device).cuda_stream)
        parts = []
        for g in reversed(groups):
            eng.set_docs([docs[i] for i in g])
            eng.run(merge_metadata=True)
            len_t, off_t, st_t = mdist.engine_rows_as_tensors(eng, device)
            hl, ho, hs = eng.rows_mum()
            assert np.array_equal(len_t.cpu().numpy().view(np.uint32), hl)
            assert np.array_equal(off_t.cpu().numpy(), ho) and np.array_equal(st_t.cpu().numpy(), hs)
            th = torch.as_tensor(mdist.DevicePointerView(eng.thresh_device_ptr(), L0 + 1), device=device)
            gathered = mdist.all_gather_partitions_device((len_t, off_t, st_t, th), dist)
            assert len(gathered) == 1 and all(t.is_cuda for t in gathered[0])
            parts.append(tuple(t.clone() for t in gathered[0]))     # the engine's buffers are reused by the next run
        parts.reverse()
        dparts = mdist.device_partitions(parts)
        merged = eng.anchor_merge(dparts, sort_like_direct=True, want_rows=False)
        order = mdist.merged_column_order(groups)
        direct = O.run([docs[i] for i in order], merge=True)
        assert merged["text"] == direct.text()
        full = eng.anchor_merge(dparts, sort_like_direct=True)
        assert full["text"] == direct.text()
        assert np.array_equal(full["thresh"], direct.thresh()[: L0 + 1])
        dl, do, ds = direct.mum_rows()
        assert np.array_equal(full["lengths"], dl) and np.array_equal(full["offsets"], do)
        assert np.array_equal(full["strands"], ds)
        # host partitions and device partitions give the same fold
        host = eng.anchor_merge([(p[0].cpu().numpy().view(np.uint32), p[1].cpu().numpy(), p[2].cpu().numpy(),
                                  p[3].cpu().numpy().view(np.uint16)) for p in parts])
        dev = eng.anchor_merge(dparts)
        for k in ("lengths", "offsets", "strands", "thresh", "text"):
            assert np.array_equal(host[k], dev[k]) if k != "text" else host[k] == dev[k]
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fold,max_text,exchange", [
    (2, "rank0", 0, "torch"), (3, "rank0", 0, "torch"), (3, "ranges", 0, "torch"), (2, "rank0", 2_000_000, "torch"),
    (2, "ranges", 2_000_000, "torch"),
    # bench.py's default: the native exchange of dist.cpp (here over the transport double of tests/fake_rccl, MUMEMTO_RCCL_LIB:
    # this box has one GPU); "ranges" = MUMEMTO_RANGE_FOLD=1, the route four ranks or more take by themselves
    (2, "rank0", 0, "native"), (3, "ranges", 0, "native"), (4, "auto", 0, "native"), (2, "ranges", 2_000_000, "native"),
    # a native exchange that fails in bench.py's trial step (FAKE_RCCL_FAIL): every rank says so and the run goes on over
    # torch.distributed, with the reason in the result line
    (3, "rank0", 0, "native-broken")])
def test_bench_multi_rank_path_on_one_gpu(world, fold, max_text, exchange):
    """bench.py's N > 1 path end to end -- torchrun, one process per rank, per-rank partition run, all-gather of the
    HBM row tables, device fold on rank 0, re-sort -- with the ranks sharing GPU 0 under gloo (this box has one GPU;
    RCCL wants one device per rank).  --check compares the merged bytes with the oracle's direct run on the union.
    max_text: the share of a rank does not fit one suffix array (whole genomes: BASELINE configs[3]) -- every rank then runs
    its share as anchor partitions + merge of its own, and what it contributes to the exchange are its merged rows and
    thresholds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world),
           "--steps", "2", "--warmup", "1", "--haps", "31", "--length", "150000", "--divergence", "0.005", "--backend", "gloo",
           "--share-device", "--check", "--exchange", exchange.split("-")[0], "--fold", fold if exchange == "torch" else "rank0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    if max_text:
        env["MMT_MAX_TEXT"] = str(max_text)
    if exchange.startswith("native"):
        if exchange == "native-broken":
            env["FAKE_RCCL_FAIL"] = "1"
        lib = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", os.path.dirname(lib)])
        env["MUMEMTO_RCCL_LIB"] = lib
        if fold != "auto":
            env["MUMEMTO_RANGE_FOLD"] = "1" if fold == "ranges" else "0"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    for _ in range(2):
        # (the port was free when it was asked for; between that and torchrun's own bind somebody else may take it: once more
        # with another one -- a launcher's matter, not the path under test)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        cmd[cmd.index("--master-port") + 1] = str(s.getsockname()[1])
        s.close()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["haplotypes"] == 31
    assert d["config"]["output_equals_cpu_oracle"] is True and d["config"]["output_bytes"] > 0
    if exchange == "native-broken":
        assert d["config"]["exchange"].startswith("torch.distributed (the native exchange failed") and "native exchange failed" in r.stderr
        assert "all-gather" in d["config"]["parallelism"]
    else:
        assert d["config"]["exchange"] == exchange


def test_bench_single_gpu_line_at_a_small_size():
    """bench.py at N = 1 on a collection the oracle can check: FASTA files -> PREFIX.mums in-process, the same job as a
    mumemto_exec process, the HBM-resident step, roofline and CPU baseline keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--haps", "12", "--length", "200000",
           "--divergence", "0.005", "--cpu-sample-bp", "200000", "--pause", "0.2", "--whole-genome", "no"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["metric"] == "input Gbp/s end-to-end" and d["value"] > 0
    assert d["config"]["output_equals_cpu_oracle"] is True
    # `value` is the process-start clock: every timed step a fresh mumemto_exec; the warm engine's figure beside it
    fp = d["fresh_processes"]
    assert [x["timed"] for x in fp["runs"]] == [False, True, True] and all(x["rc"] == 0 for x in fp["runs"])
    assert fp["output_identical_to_in_process"] is True and "fresh mumemto_exec process" in d["config"]["timed_region"].lower()
    assert abs(d["ms_per_step"] - 1e3 * sum(x["wall_s"] for x in fp["runs"][1:]) / 2) < 1e-6 and d["value_process_start"] == d["value"]
    assert d["value_in_process"] > d["value"]
    assert d["hbm_resident"]["value"] > 0 and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["cores"] == 1
    # --in-process: the timed steps are the warm engine's (what profilers that follow one process need)
    r = subprocess.run(cmd + ["--in-process", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "fresh_processes" not in d2 and d2["value_in_process"] == d2["value"] and "in-process" in d2["config"]["timed_region"]


def _sharded_worker(rank, world, port, q, wide):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "oracle"), os.path.join(root, "tests")]
    import torch
    import torch.distributed as dist
    import mumemto_amd
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth as sy
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if wide:
        os.environ.update(MMT_FORCE_WIDE="1", MMT_SCAN_RANGE="8192")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        docs = sy.pangenome(6, 30000, 0.01, seed=41, indel_rate=0.001, inversion=(3, 4000, 9000), tandem=(2, 100, 400, 3))
        eng = mumemto_amd.Engine(0)
        eng.set_docs(docs)
        out = {}
        # BASELINE configs[4] parameters (-k -1 -f 3), plain multi-MEMs, and strict multi-MUMs through the same path
        for name, kw in (("partial", dict(num_distinct=5, max_doc_freq=3, max_total_freq=18)),
                         ("mems", dict(num_distinct=2, max_doc_freq=0, max_total_freq=40)),
                         ("strict", dict(num_distinct=0, max_doc_freq=1, max_total_freq=0))):
            # the parse proper: a rank emits, scans and drops a range of the emitter's output
            out[name] = mdist.run_sharded(eng, dist, torch.device("cpu"), **kw)
            n = eng.text_length()
            st = eng.stream_stats()
            mine = eng.sort_pieces()[rank][1]
            assert len(eng.sort_pieces()) == world and sum(c for _, c in eng.sort_pieces()) == n
            # what the rank produced of the stream: its share + the left extensions of its windows, nothing else
            assert mine <= st["entries"] <= mine + st["windows"] * 8192 + 4096, (st, mine)
            assert mine <= n // world + 4096
            assert not eng.columns_kept() or n < (1 << 26)
            # the bucket-wise producer: a rank sorts, scans and drops whole bins of leading characters (batches of 3000)
            os.environ["MMT_GUIDED_BATCH"] = "3000"
            out[name + "_sort"] = mdist.run_sort_sharded(eng, dist, torch.device("cpu"), **kw)
            del os.environ["MMT_GUIDED_BATCH"]
            assert eng.producer_used() == "guided" and len(eng.sort_pieces()) == world
            assert sum(c for _, c in eng.sort_pieces()) == n
            st = eng.stream_stats()
            assert st["entries"] == eng.sort_pieces()[rank][1], "a rank of the bucket-wise producer sorts its bins and nothing else"
            # the window buffers hold two batches (+ the tail of the batch before), whatever the text length
            assert st["window_bytes"] < 10 * n // 4 and st["windows"] >= 2, st
        # a second collection for the sharded sort: a skewed alphabet (two thirds of the suffixes start with A: whole bins
        # make uneven pieces) and a run of one base that keeps a bin in one piece
        import numpy as np
        import pyoracle as O
        rng = np.random.default_rng(97)
        anc = rng.choice(np.frombuffer(b"AAAAAACGT", np.uint8), size=12000)
        anc[3000:3600] = ord("A")
        skew = []
        for d in range(5):
            h = anc.copy()
            for pos in rng.integers(0, len(h), size=40):
                h[pos] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
            skew.append([h.tobytes()])
        eng.set_docs(skew)
        got = mdist.run_sort_sharded(eng, dist, torch.device("cpu"), num_distinct=4, max_doc_freq=2, max_total_freq=0)
        pieces = [c for _, c in eng.sort_pieces()]
        out["skew_ok"] = got == O.run(skew, num_distinct=4, max_doc_freq=2, max_total_freq=0).text() and max(pieces) > min(pieces)
        q.put((rank, out))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,wide", [(2, False), (3, False), (2, True)])
def test_partial_and_mem_modes_sharded_over_ranks_equal_one_gpu(world, wide):
    """SURVEY 8(e) row 2 (BASELINE configs[4]): modes the anchor merge cannot serve run on several ranks by sharding
    the STREAM: a rank produces, scans and drops only its share -- a range of the emitter's output in the first run of
    every mode, whole bins of leading characters (the bucket-wise producer) in the second -- and no
    column is exchanged or stored; the concatenated outputs are byte for byte the oracle's single run, and what a rank
    produced of the stream is its share (checked in the workers).  The ranks share GPU 0 under gloo (this box has one GPU)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, wide)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    docs = synth.pangenome(6, 30000, 0.01, seed=41, indel_rate=0.001, inversion=(3, 4000, 9000), tandem=(2, 100, 400, 3))
    want = {"partial": O.run(docs, num_distinct=5, max_doc_freq=3, max_total_freq=18).text(),
            "mems": O.run(docs, num_distinct=2, max_doc_freq=0, max_total_freq=40).text(),
            "strict": O.run(docs).text()}
    for r in range(world):
        for name in want:
            assert got[r][name] == want[name], (r, name)
            assert got[r][name + "_sort"] == want[name], (r, name, "sharded suffix sort")
        assert got[r]["skew_ok"] is True, (r, "skewed collection through the sharded suffix sort")
    assert want["partial"].count(b"\n") > 10 and want["mems"].count(b"\n") > 10


_NATIVE_SCRIPT = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")]
if %(with_torch)r:
    import torch                      # PyTorch's own librccl.so is then in the process: the exchange must use that copy
import mumemto_amd
import pyoracle as O
from mumemto_amd import synth
docs = synth.pangenome(6, 20000, 0.01, seed=61, inversion=(2, 2000, 5000))
eng = mumemto_amd.Engine(0)
comm = mumemto_amd.Comm(eng, 0, 1, mumemto_amd.Comm.unique_id())
eng.set_docs(docs)
eng.run(merge_metadata=True)
m = comm.merge()
assert m["text"] == O.run(docs, merge=True).text() and m["n_rows"] > 5, "strict multi-MUMs through the RCCL exchange"
m2 = comm.merge(by_ranges=True)
assert m2["text"] == m["text"], "the fold by coordinate ranges (dist_merge_ranges: broadcasts, all-to-all, gather) with one rank"
# the exchange's messages with this rank as its own peer: ncclSend / ncclRecv in one group, all-gather, broadcast
lb = comm.loopback()
assert lb["different"] == 0 and lb["rows"] == m["n_rows"] and lb["thresholds"] == len(docs[0][0]) + 1, lb
assert lb["pieces"] >= (8 if os.environ.get("MUMEMTO_RCCL_CHUNK") else 4), lb
eng.set_scan_shard(0, 1)
eng.run(num_distinct=5, max_doc_freq=3, max_total_freq=18)
assert comm.gather_text() == O.run(docs, num_distinct=5, max_doc_freq=3, max_total_freq=18).text()
# the bucket-wise producer with its one share
eng.set_producer("guided")
eng.run(num_distinct=5, max_doc_freq=3, max_total_freq=18)
eng.set_producer("auto")
assert eng.sort_pieces() == [(0, eng.text_length())] and eng.producer_used() == "guided"
assert comm.gather_text() == O.run(docs, num_distinct=5, max_doc_freq=3, max_total_freq=18).text()
comm.close(); eng.close()
print("NATIVE_EXCHANGE_OK")
"""


@pytest.mark.parametrize("with_torch", [False, True])
def test_c_abi_exchange_over_rccl_world_size_one(with_torch):
    """mmt_comm_* / mmt_dist_* (dist.cpp: ncclCommInitRank, grouped ncclBroadcast of the row tables, fold on rank 0,
    gather of the sharded modes' bytes) with the one rank a one-GPU box allows -- in a process of its own, once without
    PyTorch (RCCL from /opt/rocm) and once with PyTorch's copy already loaded."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if not with_torch:
        env["MUMEMTO_NO_TORCH"] = "1"
        env["MUMEMTO_RCCL_CHUNK"] = "4099"          # (every message of the self-test in pieces of 4099 elements)
    r = subprocess.run([sys.executable, "-c", _NATIVE_SCRIPT % dict(root=root, with_torch=with_torch)], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NATIVE_EXCHANGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("on_device", [False, True])
def test_fold_by_coordinate_ranges_equals_the_whole_fold(on_device):
    """merge.cpp anchor_merge_by_ranges: slice r of N near-equal slices of the anchor folds from the rows that start in
    [lo - margin, hi) and the thresholds of that range alone (margin = (partitions - 1) x the longest row + 1) -- what rank r
    of N does in dist_merge_ranges, here one slice after the other on one device.  Rows, strands, thresholds and the bytes
    of the output equal the fold of the whole anchor for 1 .. 13 slices (slices shorter than the margin included), for host
    and for device partitions, and the re-sorted result equals the direct run."""
    import mumemto_amd
    docs = synth.pangenome(12, 40000, 0.01, seed=77, inversion=(3, 3000, 9000), indel_rate=0.0005)
    groups = [[0, 1, 2, 3], [0, 4, 5], [0, 6, 7, 8], [0, 9, 10, 11]]
    L0 = len(docs[0][0])
    eng = mumemto_amd.Engine(0)
    try:
        parts = []
        for g in reversed(groups):                      # (the last run = partition 0: the engine keeps the anchor ranks)
            eng.set_docs([docs[i] for i in g])
            eng.run(merge_metadata=True)
            l, o, st = eng.rows_mum()
            parts.append((l.copy(), o.copy(), st.copy(), eng.thresholds()[: L0 + 1].copy()))
        parts.reverse()
        if on_device:
            import torch
            from mumemto_amd import dist as mdist
            dev = torch.device("cuda", 0)
            tens = [(torch.from_numpy(p[0].view(np.int32)).to(dev), torch.from_numpy(p[1]).to(dev), torch.from_numpy(p[2]).to(dev),
                     torch.from_numpy(p[3].view(np.int16)).to(dev)) for p in parts]
            use = mdist.device_partitions(tens)
        else:
            use = parts
        whole = eng.anchor_merge(use)
        assert len(whole["lengths"]) > 50
        for slices in (1, 2, 3, 5, 8, 13, 400):
            got = eng.anchor_merge(use, slices=slices)
            for k in ("lengths", "offsets", "strands", "thresh"):
                assert np.array_equal(got[k], whole[k]), (slices, k)
            assert got["text"] == whole["text"], slices
        order = [i for g in groups for i in (g if g is groups[0] else g[1:])]
        direct = O.run([docs[i] for i in order], merge=True)
        assert eng.anchor_merge(use, sort_like_direct=True, slices=5)["text"] == direct.text()
    finally:
        eng.close()
