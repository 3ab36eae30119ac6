"""GPU tests of the CLI contract: FASTA files in, PREFIX.{mums,mems,bumbl,lengths,
athresh,thresh,thresh_rev,sa,lcp,bwt} out, byte-compared with the oracle's writers;
anchor_merge tool against the real reference binary's golden output."""
import glob
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import build, synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(build.LIB), "..", "bin")
HERE = os.path.dirname(os.path.abspath(__file__))


def write_inputs(tmp_path, docs):
    paths = []
    for i, d in enumerate(docs):
        if i == 1:   # two records + lowercase + 60 columns
            rec = d[0]
            recs = [rec[: len(rec) // 3].lower(), rec[len(rec) // 3:]]
            p = tmp_path / ("g%d.fasta" % i)
            synth.write_fasta(str(p), recs, names=["a", "b"], width=60)
        elif i == 2:
            p = tmp_path / ("g%d.fa.gz" % i)
            raw = tmp_path / "tmp.fa"
            synth.write_fasta(str(raw), d)
            with gzip.open(p, "wb") as f:
                f.write(raw.read_bytes())
        else:
            p = tmp_path / ("g%d.fa" % i)
            synth.write_fasta(str(p), d)
        paths.append(str(p))
    return paths


def cli(args, cwd):
    r = subprocess.run([os.path.join(BIN, "mumemto_exec")] + args, cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("cli")
    docs = synth.pangenome(6, 30000, 0.01, seed=12, indel_rate=0.002, inversion=(3, 5000, 9000))
    return tmp, docs, write_inputs(tmp, docs)


def test_default_partial_mem_and_norevcomp(inputs):
    tmp, docs, paths = inputs
    N = len(docs)
    for name, args, kw, ext in [
        ("def", [], dict(), "mums"),
        ("k1", ["-k", "-1"], dict(num_distinct=N - 1), "mums"),
        ("mem", ["-k", "-1", "-f", "3"], dict(num_distinct=N - 1, max_doc_freq=3, max_total_freq=3 * N), "mems"),
        ("nor", ["-r"], dict(revcomp=False), "mums"),
        ("F", ["-f", "0", "-k", "2", "-F", "9", "-l", "25"], dict(num_distinct=2, max_doc_freq=0, max_total_freq=9,
                                                                 min_len=25), "mems"),
        ("g", ["-g", "-w", "12", "-m", "50"], dict(), "mums"),
    ]:
        cli(["-o", str(tmp / name)] + args + paths, tmp)
        if "num_distinct" not in kw and "max_doc_freq" not in kw:
            kw = dict(kw, max_total_freq=N)
        elif kw.get("max_doc_freq", 1) == 1 and "max_total_freq" not in kw:
            kw = dict(kw, max_total_freq=N)
        want = O.run(docs, **kw)
        assert (tmp / (name + "." + ext)).read_bytes() == want.text(), name


def test_binary_and_merge_metadata_outputs(inputs):
    tmp, docs, paths = inputs
    cli(["-o", str(tmp / "b"), "-b"] + paths, tmp)
    want = O.run(docs, max_total_freq=len(docs), merge=True)
    assert (tmp / "b.bumbl").read_bytes() == want.bumbl()
    cli(["-o", str(tmp / "an"), "-n"] + paths, tmp)
    assert (tmp / "an.mums").read_bytes() == want.text()
    L0 = len(docs[0][0])
    assert (tmp / "an.athresh").read_bytes() == want.thresh()[: L0 + 1].tobytes()
    cli(["-o", str(tmp / "st"), "-M"] + paths, tmp)
    assert (tmp / "st.thresh").read_bytes() == want.thresh_file(False).tobytes()
    assert (tmp / "st.thresh_rev").read_bytes() == want.thresh_file(True).tobytes()


def test_arrays_out(inputs):
    tmp, docs, paths = inputs
    cli(["-o", str(tmp / "arr"), "-A"] + paths, tmp)
    text, _ = O.build_text(docs, True)
    sa, lcp, bwt = O.build_stream(text)
    n1 = len(sa)

    def u40(path):
        raw = np.fromfile(path, np.uint8).reshape(-1, 5).astype(np.uint64)
        return sum(raw[:, k] << np.uint64(8 * k) for k in range(5))
    assert np.array_equal(u40(tmp / "arr.sa"), sa.astype(np.uint64)) and len(sa) == n1
    assert np.array_equal(u40(tmp / "arr.lcp"), lcp.astype(np.uint64))
    got_bwt = np.fromfile(tmp / "arr.bwt", np.uint8)
    assert np.array_equal(got_bwt[1:], bwt[1:]) and got_bwt[0] == text[-1]


def test_only_parse_writes_reference_compatible_files(tmp_path):
    # -P with the reference's default w/p: same bytes as the real reference parser (golden fixture)
    G = os.path.join(HERE, "golden", "newscan", "three_docs_w10_p100")
    docs, cur = [], []
    for line in open(os.path.join(G, "input.txt"), "rb").read().split(b"\n"):
        if line.startswith(b"F $"):
            if cur:
                docs.append(cur)
                cur = []
        elif line.startswith(b"F "):
            cur.append(line[2:])
    paths = []
    for i, d in enumerate(docs):
        p = tmp_path / ("d%d.fa" % i)
        synth.write_fasta(str(p), d)
        paths.append(str(p))
    cli(["-o", str(tmp_path / "pp"), "-P"] + paths, tmp_path)
    assert (tmp_path / "pp.dict").read_bytes() == open(os.path.join(G, "out.dict"), "rb").read()
    assert (tmp_path / "pp.parse").read_bytes() == open(os.path.join(G, "out.parse"), "rb").read()
    assert not (tmp_path / "pp.mums").exists()
    cli(["-o", str(tmp_path / "kk"), "-K"] + paths, tmp_path)
    assert (tmp_path / "kk.dict").read_bytes() == open(os.path.join(G, "out.dict"), "rb").read()
    assert (tmp_path / "kk.mums").exists()


def test_anchor_merge_tool_matches_reference_binary(tmp_path):
    G = os.path.join(HERE, "golden", "anchor_merge")
    for case in sorted(os.listdir(G)):
        work = tmp_path / case
        shutil.copytree(os.path.join(G, case), work)
        parts = sorted(glob.glob(str(work / "p*.mums")))
        r = subprocess.run([os.path.join(BIN, "anchor_merge")] + parts + ["-o", str(work / "out")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (work / "out.mums").read_bytes() == (work / "merged.mums").read_bytes()
        assert (work / "out.athresh").read_bytes() == (work / "merged.athresh").read_bytes()
        # .bumbl in, .bumbl out
        r = subprocess.run([os.path.join(BIN, "anchor_merge")] + parts + ["-o", str(work / "outb.bumbl")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert os.path.getsize(work / "outb.bumbl") > 18


def test_checkpoints_from_parse_and_arrays_in(inputs, tmp_path):
    """-K / -A write the stage checkpoints, -p / -a start from them (src/pfp_mum.cpp:97-111, :122-124).  -p must
    reproduce the direct run byte for byte; -a scans the first |T| stream entries only, exactly like the reference's
    file reader (include/read_arrays.hpp:86-104), so it is compared with the oracle's scan of that truncated stream."""
    tmp, docs, paths = inputs
    cli(["-o", str(tmp_path / "ck"), "-K", "-A", "-w", "8", "-m", "40"] + paths, tmp_path)
    direct = (tmp_path / "ck.mums").read_bytes()
    assert direct == O.run(docs).text()
    for extra, want in ([], None), (["-k", "-1", "-f", "3"], None):
        cli(["-o", str(tmp_path / "fromp"), "-p", str(tmp_path / "ck"), "-w", "8"] + extra, tmp_path)
        out = (tmp_path / ("fromp.mums" if not extra else "fromp.mems")).read_bytes()
        nd, f, F = O.cli_params(len(docs), *([0, 1, 0] if not extra else [-1, 3, 0]))
        assert out == O.run(docs, num_distinct=nd, max_doc_freq=f, max_total_freq=F).text()
    assert not (tmp_path / "fromp.lengths").exists()          # the checkpoint runs only read PREFIX.lengths
    text, doc_start = O.build_text(docs, True)
    sa, lcp, bwt = O.build_stream(text)
    n = len(text)
    cli(["-o", str(tmp_path / "froma"), "-a", str(tmp_path / "ck")], tmp_path)
    want = O.scan(sa[:n], lcp[:n], bwt[:n], doc_start)
    assert (tmp_path / "froma.mums").read_bytes() == want.text()
    cli(["-o", str(tmp_path / "fromam"), "-a", str(tmp_path / "ck"), "-M", "-n"], tmp_path)
    wantm = O.scan(sa[:n], lcp[:n], bwt[:n], doc_start, merge=True)
    assert (tmp_path / "fromam.mums").read_bytes() == wantm.text()
    L0 = len(b"".join(docs[0]))
    assert (tmp_path / "fromam.athresh").read_bytes() == wantm.thresh()[: L0 + 1].tobytes()


def test_engine_accepts_text_and_stream():
    """The same checkpoints through the device-resident ABI (mmt_engine_set_text_host / _set_stream_host)."""
    import mumemto_amd
    docs = synth.pangenome(5, 20000, 0.01, seed=41, inversion=(2, 3000, 6000))
    lens = [len(b"".join(d)) for d in docs]
    for revcomp in (True, False):
        text, doc_start = O.build_text(docs, revcomp)
        sa, lcp, bwt = O.build_stream(text)
        want = O.run(docs, revcomp=revcomp)
        eng = mumemto_amd.Engine(0)
        eng.set_text(bytes(text), lens, use_revcomp=revcomp)
        eng.run(use_revcomp=revcomp)
        assert eng.output_text() == want.text()
        eng.set_stream(sa[1:], lcp[1:], bwt[1:], lens, use_revcomp=revcomp)
        eng.run(use_revcomp=revcomp)
        assert eng.output_text() == want.text()
        eng.run(use_revcomp=revcomp, num_distinct=3, max_doc_freq=2)
        assert eng.output_text() == O.run(docs, revcomp=revcomp, num_distinct=3, max_doc_freq=2).text()
        with pytest.raises(mumemto_amd.MumemtoError):
            eng.run(use_revcomp=not revcomp)
        eng.set_docs(docs)                                   # back to the normal path
        eng.run(use_revcomp=revcomp)
        assert eng.output_text() == want.text()
        eng.close()


def test_gpus_option_runs_one_process_per_gpu_and_the_rccl_exchange(inputs):
    """`mumemto_exec --gpus N`: a launcher + one rank process per GPU, the C++ exchange of dist.cpp (RCCL) between them.
    This box has one GPU, so the launcher is forced for N = 1 (MUMEMTO_FORCE_RANKS): the rank process, the communicator
    id left in a file, the collective calls, the fold and the writers all run; what a second rank adds are more
    broadcasts of the same kind (tests/test_gpu_dist.py runs the fold over several partitions)."""
    tmp, docs, paths = inputs
    N = len(docs)
    env = dict(os.environ, MUMEMTO_FORCE_RANKS="1")
    for name, args, kw, ext in [
        ("r_def", [], dict(max_total_freq=N), "mums"),
        ("r_n", ["-n", "-l", "15"], dict(max_total_freq=N, min_len=15, merge=True), "mums"),
        ("r_mem", ["-k", "-1", "-f", "3"], dict(num_distinct=N - 1, max_doc_freq=3, max_total_freq=3 * N), "mems"),
    ]:
        r = subprocess.run([os.path.join(BIN, "mumemto_exec"), "--gpus", "1", "-o", str(tmp / name)] + args + paths, cwd=tmp,
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        want = O.run(docs, **kw)
        assert (tmp / (name + "." + ext)).read_bytes() == want.text(), name
        if "-n" in args:
            got = np.frombuffer((tmp / (name + ".athresh")).read_bytes(), np.uint16)
            assert np.array_equal(got, want.thresh()[: len(docs[0][0]) + 1])
        assert not glob.glob(str(tmp / (name + ".comm.*"))) and not glob.glob(str(tmp / (name + ".rank*")))
    # the sharded suffix sort through the command line (one piece here: the callback, the in-place broadcasts)
    r = subprocess.run([os.path.join(BIN, "mumemto_exec"), "--gpus", "1", "-o", str(tmp / "r_sort"), "-k", "-1", "-f", "3"] + paths,
                       cwd=tmp, capture_output=True, text=True, env=dict(env, MUMEMTO_SORT_SHARD="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp / "r_sort.mems").read_bytes() == (tmp / "r_mem.mems").read_bytes()
    # the lengths file of the rank processes is the one a single process writes
    cli(["-o", str(tmp / "r_one")] + paths, tmp)
    assert (tmp / "r_def.lengths").read_bytes() == (tmp / "r_one.lengths").read_bytes()
    # a request the launcher cannot serve fails with a message, not a hang
    r = subprocess.run([os.path.join(BIN, "mumemto_exec"), "--gpus", "9", "-o", str(tmp / "r_bad")] + paths, cwd=tmp,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "fewer documents" in r.stderr


REF_MERGE = os.path.join(HERE, "..", "oracle", "_ref", "anchor_merge")


@pytest.mark.skipif(not os.path.exists(REF_MERGE), reason="the reference's anchor_merge was not built (oracle/Makefile ref)")
@pytest.mark.parametrize("grouping", [[[1, 2], [3, 4, 5]], [[1], [2, 3], [4, 5]], [[1, 2, 3], [4, 5]]])
def test_real_reference_anchor_merge_eats_gpu_partitions_at_megabase_size(tmp_path, grouping):
    """The REAL reference tool (src/merge_candidates.cpp compiled unmodified) folds PREFIX.mums / PREFIX.athresh written
    by the GPU command line for anchor partitions of 6 x 1.2 Mbp, in the three groupings of the golden fixtures; its rows
    re-sorted by match string are the bytes of the GPU's direct run on all six documents, its merged .athresh the direct
    run's (up to the reference's end-of-stream quirk: at most one row per partition, and thresholds where
    the match of all documents is shorter than -l)."""
    from mumsfile import format_mums, parse_mums
    docs = synth.pangenome(6, 1_200_000, 0.004, seed=77, indel_rate=0.0005, inversion=(2, 100_000, 160_000))
    paths = []
    for i, d in enumerate(docs):
        p = tmp_path / ("m%d.fa" % i)
        synth.write_fasta(str(p), d)
        paths.append(str(p))
    cli(["-o", str(tmp_path / "direct"), "-M", "-n"] + paths, tmp_path)
    direct = (tmp_path / "direct.mums").read_bytes()
    direct_th = np.fromfile(tmp_path / "direct.athresh", np.uint16)
    assert direct.count(b"\n") > 1000
    parts = []
    for k, g in enumerate(grouping):
        cli(["-o", str(tmp_path / ("p%d" % k)), "-M", "-n", paths[0]] + [paths[i] for i in g], tmp_path)
        parts.append(str(tmp_path / ("p%d.mums" % k)))
    r = subprocess.run([REF_MERGE] + parts + ["-o", str(tmp_path / "merged")], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    L, off, st = parse_mums((tmp_path / "merged.mums").read_bytes())
    anchor = b"".join(docs[0]).upper()
    order = sorted(range(len(L)), key=lambda i: anchor[off[i, 0]: off[i, 0] + L[i]])
    resorted = format_mums(L[order], off[order], st[order])
    if resorted != direct:
        a, b = set(direct.split(b"\n")), set(resorted.split(b"\n"))
        assert len(b - a) == 0 and len(a - b) <= len(grouping), (len(a - b), len(b - a))
    merged_th = np.fromfile(tmp_path / "merged.athresh", np.uint16)
    differ = np.nonzero(merged_th != direct_th)[0]
    assert len(differ) <= 5 and np.all(direct_th[differ] == 0)


def test_streamed_input_documents_read_when_the_device_asks(inputs):
    """MUMEMTO_STREAM_INPUT=1: what mumemto_exec does with a collection that does not fit the host as bytes (94 whole genomes:
    287 GB; the reference streams its files through the parser, src/ref_builder.cpp:211-314) -- every file measured once, then
    read again one document at a time when the engine asks (Engine::run_supplied), rows written window by window.  Same
    files as the resident route, byte for byte: .mums / .mems / .bumbl / .lengths / .athresh / .thresh."""
    tmp, docs, paths = inputs
    N = len(docs)
    env = dict(os.environ, MUMEMTO_STREAM_INPUT="1")
    for name, args, exts in [
        ("s_def", [], ["mums", "lengths"]),
        ("s_mem", ["-k", "-1", "-f", "3"], ["mems", "lengths"]),
        ("s_n", ["-n"], ["mums", "athresh"]),
        ("s_M", ["-M"], ["mums", "thresh", "thresh_rev"]),
        ("s_b", ["-b"], ["bumbl"]),
        ("s_packed", ["-k", "-2"], ["mums"]),
    ]:
        e = dict(env, MMT_PACKED_TEXT="1", MUMEMTO_PRODUCER="guided") if name == "s_packed" else env
        r = subprocess.run([os.path.join(BIN, "mumemto_exec"), "-o", str(tmp / name)] + args + paths, cwd=tmp,
                           capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        assert "measured %d files" % N in r.stderr
        cli(["-o", str(tmp / (name + "_resident"))] + args + paths, tmp)
        for ext in exts:
            a, b = (tmp / (name + "." + ext)).read_bytes(), (tmp / (name + "_resident." + ext)).read_bytes()
            assert a == b and len(a) > 0, (name, ext)
    # options that need the collection on the host say so
    r = subprocess.run([os.path.join(BIN, "mumemto_exec"), "-o", str(tmp / "s_K"), "-K"] + paths, cwd=tmp, capture_output=True,
                       text=True, env=env)
    assert r.returncode != 0 and "streamed" in r.stderr
