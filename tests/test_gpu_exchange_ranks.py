"""GPU: the native exchange of dist.cpp (mmt_comm_* / mmt_dist_*) with MORE THAN ONE RANK on a one-GPU box.

RCCL wants one device per rank, so the glue of dist.cpp -- which table goes to whom, `thresh + base[r]`, `hi[r] - base[r]`
against the receiver's span, counts of the all-to-all, the order the pieces are gathered in -- had only ever run with
world = 1, where every loop is empty.  Here the ten RCCL symbols dist.cpp binds come from the transport double of
tests/fake_rccl (MUMEMTO_RCCL_LIB; ranks = processes sharing GPU 0, messages staged through /dev/shm, a receive whose
size differs from the message FAILS), and everything above the transport is the product code: rank r runs {anchor} + its
share with merge metadata, `mmt_dist_merge` (rank 0 folds) and `mmt_dist_merge_ranges` (every rank folds its slice of the
anchor: all-to-all of row and threshold slices, pieces to rank 0) must give the bytes of the oracle's direct run on the
union; `mmt_dist_gather_text` must give the single-GPU bytes of the sharded modes; `mumemto_exec --gpus N` runs the same
through the command line."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth
from mumemto_amd import dist as mdist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def fake_lib():
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "fake_rccl")])
    return FAKE


_WORKER = r"""
import os, sys, time
root = %(root)r
sys.path[:0] = [root, os.path.join(root, "oracle"), os.path.join(root, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd import dist as mdist
rank, world, idfile, outdir, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
docs = synth.pangenome(%(haps)d, %(length)d, 0.01, seed=%(seed)d, inversion=(2, 2000, 5000), indel_rate=0.0005)
eng = mumemto_amd.Engine(0)
if rank == 0:
    uid = mumemto_amd.Comm.unique_id()
    with open(idfile + ".tmp", "wb") as f: f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
comm = mumemto_amd.Comm(eng, rank, world, uid)
if what in ("rank0", "ranges", "auto"):
    mine = mdist.partition_docs(len(docs), world)[rank]
    if %(max_text)d:
        os.environ["MMT_MAX_TEXT"] = str(%(max_text)d)        # the share runs as anchor partitions inside the rank
    eng.run_partitioned([docs[i] for i in mine], merge_metadata=True)
    if what == "auto":
        m = comm.merge()
    else:
        m = comm.merge(by_ranges=(what == "ranges"))
    assert (m is not None) == (rank == 0)
    if rank == 0:
        open(os.path.join(outdir, "merged.mums"), "wb").write(m["text"])
else:
    kw = dict(num_distinct=%(haps)d - 1, max_doc_freq=3, max_total_freq=3 * %(haps)d)
    if what == "sharded_sort":
        eng.set_producer("guided")
        os.environ["MMT_GUIDED_BATCH"] = "4000"
    eng.set_docs(docs)
    eng.set_scan_shard(rank, world)
    eng.run(**kw)
    text = comm.gather_text()
    assert (len(text) > 0) == (rank == 0)
    if rank == 0:
        open(os.path.join(outdir, "gathered.mems"), "wb").write(text)
comm.close(); eng.close()
print("RANK_OK", rank)
"""


def run_ranks(world, what, haps=9, length=30000, seed=71, max_text=0, env_extra=None):
    lib = fake_lib()
    with tempfile.TemporaryDirectory(prefix="ranks_", dir="/dev/shm") as d:
        script = os.path.join(d, "worker.py")
        with open(script, "w") as f:
            f.write(_WORKER % dict(root=ROOT, haps=haps, length=length, seed=seed, max_text=max_text))
        env = dict(os.environ, MUMEMTO_RCCL_LIB=lib, MUMEMTO_NO_TORCH="1")
        env.update(env_extra or {})
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), os.path.join(d, "id"), d, what], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=600))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        for r, (p, (so, se)) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and "RANK_OK" in so, "rank %d:\n%s\n%s" % (r, so[-2000:], se[-4000:])
        name = "merged.mums" if what in ("rank0", "ranges", "auto") else "gathered.mems"
        with open(os.path.join(d, name), "rb") as f:
            return f.read()


def direct_bytes(haps, length, seed, world):
    docs = synth.pangenome(haps, length, 0.01, seed=seed, inversion=(2, 2000, 5000), indel_rate=0.0005)
    order = mdist.merged_column_order(mdist.partition_docs(haps, world))
    return O.run([docs[i] for i in order], merge=True).text()


@pytest.mark.parametrize("world,what", [(2, "rank0"), (3, "rank0"), (2, "ranges"), (3, "ranges"), (4, "ranges"), (8, "ranges"),
                                        (4, "auto"), (5, "rank0")])
def test_exchange_and_fold_over_several_ranks(world, what):
    got = run_ranks(world, what)
    want = direct_bytes(9, 30000, 71, world)
    assert want.count(b"\n") > 20
    assert got == want


@pytest.mark.parametrize("world,what", [(3, "rank0"), (4, "ranges")])
def test_messages_travel_in_pieces_when_their_counts_are_large(world, what):
    """dist.cpp cuts every message into pieces of at most 2^29 bytes (2^30 elements until round 6) (a threshold column over a 3.05 Gbp anchor is 3.05 G
    elements): MUMEMTO_RCCL_CHUNK = 997 makes every table of this small collection travel in dozens of pieces, sender and
    receiver cutting the same way -- the merged bytes are still the oracle's."""
    got = run_ranks(world, what, env_extra={"MUMEMTO_RCCL_CHUNK": "997"})
    assert got == direct_bytes(9, 30000, 71, world)


def test_route_is_rank_zeros_decision():
    """The ranks follow rank 0's choice of the fold even when their environments disagree (a rank that read another
    MUMEMTO_RANGE_FOLD used to enter another collective and hang): the workers get different values by rank."""
    lib = fake_lib()
    got = run_ranks(3, "auto", env_extra={"MUMEMTO_RANGE_FOLD": "1"})
    assert got == direct_bytes(9, 30000, 71, 3)
    # (per-rank environments: the launcher below sets the variable for rank 1 only)
    with tempfile.TemporaryDirectory(prefix="ranks_", dir="/dev/shm") as d:
        script = os.path.join(d, "worker.py")
        with open(script, "w") as f:
            f.write(_WORKER % dict(root=ROOT, haps=9, length=30000, seed=71, max_text=0))
        procs = []
        for r in range(3):
            env = dict(os.environ, MUMEMTO_RCCL_LIB=lib, MUMEMTO_NO_TORCH="1")
            if r == 1:
                env["MUMEMTO_RANGE_FOLD"] = "1"
            procs.append(subprocess.Popen([sys.executable, script, str(r), "3", os.path.join(d, "id"), d, "auto"], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for r, p in enumerate(procs):
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0, "rank %d:\n%s\n%s" % (r, so[-2000:], se[-4000:])
        assert open(os.path.join(d, "merged.mums"), "rb").read() == direct_bytes(9, 30000, 71, 3)


def test_a_share_that_ran_as_partitions_inside_the_rank_goes_through_the_exchange():
    """BASELINE configs[3] with a share beyond one suffix array: the rank merges its own partitions first and contributes
    the merged rows and thresholds."""
    got = run_ranks(2, "ranges", haps=11, length=20000, seed=73, max_text=150000)
    assert got == direct_bytes(11, 20000, 73, 2)
    got = run_ranks(2, "rank0", haps=11, length=20000, seed=73, max_text=150000)
    assert got == direct_bytes(11, 20000, 73, 2)


@pytest.mark.parametrize("world,what", [(2, "sharded"), (3, "sharded"), (3, "sharded_sort")])
def test_gather_of_the_sharded_modes_over_several_ranks(world, what):
    got = run_ranks(world, what, haps=6, length=20000, seed=75)
    docs = synth.pangenome(6, 20000, 0.01, seed=75, inversion=(2, 2000, 5000), indel_rate=0.0005)
    want = O.run(docs, num_distinct=5, max_doc_freq=3, max_total_freq=18).text()
    assert want.count(b"\n") > 10
    assert got == want


@pytest.mark.parametrize("gpus,mode", [(2, "strict"), (4, "strict"), (3, "partial"), (3, "strict-streamed")])
def test_mumemto_exec_gpus_n_on_one_gpu(gpus, mode, tmp_path):
    """`mumemto_exec --gpus N`: the launcher, one process per rank (sharing GPU 0: MUMEMTO_SHARE_DEVICE), the exchange,
    PREFIX.mums / .lengths / .athresh written by rank 0."""
    exe = os.path.join(ROOT, "mumemto_amd", "bin", "mumemto_exec")
    docs = synth.pangenome(9, 30000, 0.01, seed=79, inversion=(3, 2000, 5000))
    paths = []
    for i, d in enumerate(docs):
        p = str(tmp_path / ("h%02d.fa" % i))
        synth.write_fasta(p, d)
        paths.append(p)
    env = dict(os.environ, MUMEMTO_RCCL_LIB=fake_lib(), MUMEMTO_SHARE_DEVICE="1")
    if mode == "strict-streamed":      # every rank reads its share (anchor + block) one document at a time as its engine asks
        env["MUMEMTO_STREAM_INPUT"] = "1"
        mode = "strict"
    out = str(tmp_path / "out")
    args = [exe, "-o", out, "--gpus", str(gpus)] + (["-n"] if mode == "strict" else ["-k", "-1", "-f", "3"]) + paths
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if mode == "strict":
        order = mdist.merged_column_order(mdist.partition_docs(9, gpus))
        want = O.run([docs[i] for i in order], merge=True)
        assert open(out + ".mums", "rb").read() == want.text()
        assert np.array_equal(np.fromfile(out + ".athresh", np.uint16), want.thresh()[: len(docs[0][0]) + 1])
        lines = open(out + ".lengths").read().splitlines()
        assert len(lines) == 2 * 9
    else:
        want = O.run(docs, num_distinct=8, max_doc_freq=3, max_total_freq=27)
        assert open(out + ".mems", "rb").read() == want.text()


@pytest.mark.parametrize("streamed", [0, 1])
def test_ranks_of_a_sharded_run_write_pieces_and_nothing_is_gathered(streamed, tmp_path):
    """MUMEMTO_RANK_PIECES=1 (automatic when the collection has to be streamed): every rank of a `-k / -f` run writes the rows of
    its share of the stream window by window to PREFIX.rankR.mems and the launcher joins the pieces in rank order -- no
    communicator, no RCCL library, nothing in HBM or on the links (a rank of BASELINE configs[4] writes 66 GB).  With
    MUMEMTO_STREAM_INPUT=1 the ranks also read their documents one at a time as their engines ask."""
    exe = os.path.join(ROOT, "mumemto_amd", "bin", "mumemto_exec")
    docs = synth.pangenome(9, 30000, 0.01, seed=79, inversion=(3, 2000, 5000))
    paths = []
    for i, d in enumerate(docs):
        p = str(tmp_path / ("h%02d.fa" % i))
        synth.write_fasta(p, d)
        paths.append(p)
    env = dict(os.environ, MUMEMTO_RCCL_LIB="/nonexistent/librccl.so", MUMEMTO_SHARE_DEVICE="1", MUMEMTO_RANK_PIECES="1",
               MUMEMTO_STREAM_INPUT=str(streamed), MMT_SCAN_RANGE="16384")
    for name, args, kw, ext in (("mem", ["-k", "-1", "-f", "3"], dict(num_distinct=8, max_doc_freq=3, max_total_freq=27), "mems"),
                                ("par", ["-k", "-2"], dict(num_distinct=7, max_total_freq=9), "mums")):
        out = str(tmp_path / name)
        r = subprocess.run([exe, "-o", out, "--gpus", "3"] + args + paths, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        want = O.run(docs, **kw).text()
        assert open(out + "." + ext, "rb").read() == want and want.count(b"\n") > 10
        assert not [f for f in os.listdir(tmp_path) if ".rank" in f], "pieces left behind"
        assert len(open(out + ".lengths").read().splitlines()) == 2 * 9


_P2P = r"""
import ctypes as C, os, sys, time
import torch
lib = C.CDLL(%(fake)r)
rank, idfile, order = int(sys.argv[1]), sys.argv[2], sys.argv[3]
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = Uid()
if rank == 0:
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    with open(idfile + ".tmp", "wb") as f: f.write(bytes(uid))
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    C.memmove(C.byref(uid), open(idfile, "rb").read(), 128)
comm = C.c_void_p()
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert lib.ncclCommInitRank(C.byref(comm), 2, uid, rank) == 0
for f in (lib.ncclSend, lib.ncclRecv):
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.ncclGetErrorString.restype = C.c_char_p
mine = torch.full((1000,), rank + 1, dtype=torch.int32, device="cuda:0")
got = torch.zeros(1000, dtype=torch.int32, device="cuda:0")
peer = 1 - rank
U32 = 3                                      # ncclUint32
def send(): return lib.ncclSend(mine.data_ptr(), 1000, U32, peer, comm, None)
def recv(): return lib.ncclRecv(got.data_ptr(), 1000, U32, peer, comm, None)
if order == "grouped":                      # both ranks: send first, then receive -- inside one group: progresses together
    lib.ncclGroupStart(); rc = send() or recv(); rc = lib.ncclGroupEnd() or rc
elif order == "ordered":                    # rank 0 sends then receives, rank 1 receives then sends: matched
    rc = (send() or recv()) if rank == 0 else (recv() or send())
else:                                       # "crossed": both send before they receive, outside a group: a deadlock on real links
    rc = send() or recv()
if rc:
    print("ERROR", lib.ncclGetErrorString(rc).decode()); sys.exit(7)
torch.cuda.synchronize()
assert int(got[0]) == peer + 1 and int(got[-1]) == peer + 1
print("OK")
"""


@pytest.mark.parametrize("order", ["grouped", "ordered", "crossed"])
def test_the_transport_double_keeps_rendezvous_semantics(order, tmp_path):
    """A send is complete only when its receive has run -- in a group the operations progress together, outside a group two
    ranks that send to each other before they receive hang on real links: the double must fail there ("probable deadlock"),
    or ordering bugs of the exchange could never show up before a multi-GPU node does."""
    script = tmp_path / "p2p.py"
    script.write_text(_P2P % dict(fake=fake_lib()))
    idfile = str(tmp_path / "id")
    env = dict(os.environ, FAKE_RCCL_TIMEOUT="4")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), idfile, order], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    if order == "crossed":
        assert all(p.returncode == 7 for p in procs), [o[1][-300:] for o in outs]
        assert all(b"probable deadlock" in o[0] for o in outs)
    else:
        assert all(p.returncode == 0 and b"OK" in o[0] for p, o in zip(procs, outs)), [o[1][-400:] for o in outs]
