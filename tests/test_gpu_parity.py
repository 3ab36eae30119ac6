"""GPU parity tests: the HIP path (through the C ABI of libmumemto.so) against
the oracle, bit-exact, on the same seeded inputs -- stage by stage (text, SA,
LCP, BWT) and end to end (.mums/.mems bytes, row arrays, thresholds)."""
import json
import os

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth
from conftest import producer_is

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["pfp", "direct", "guided", "expand", "expand12"])
def engine(request):
    """Every case through all SA/LCP/BWT producers: prefix-free parsing with the production window (w 6, p 16; what the
    automatic choice takes from five documents on), the direct suffix sort (its choice for fewer), the parse without
    the suffix array of its dictionary (what whole-genome partitions of little redundancy fall back to), here in batches of
    20000 suffixes, and the same with expansion (what a rank's share of whole genomes takes: one representative per distinct
    phrase suffix is sorted, the emitter expands it; batches of 6000 representatives; "expand12": with the 12-byte
    occurrence records of texts whose positions and parse ranks exceed 64 bits together)."""
    import mumemto_amd
    e = mumemto_amd.Engine(0)
    if request.param == "pfp":
        e.set_producer("pfp", 6, 16)
    elif request.param == "guided":
        e.set_producer("guided", 6, 16)
        os.environ["MMT_GUIDED_BATCH"] = "20000"
    elif request.param in ("expand", "expand12"):
        e.set_producer("expand", 6, 16)
        os.environ["MMT_GUIDED_BATCH"] = "6000"
        if request.param == "expand12":
            os.environ["MMT_OCC_REC12"] = "1"
    else:
        e.set_producer("direct")
    yield e
    os.environ.pop("MMT_GUIDED_BATCH", None)
    os.environ.pop("MMT_OCC_REC12", None)
    e.close()


@pytest.mark.skipif(os.environ.get("MMT_FORCE_WIDE") is not None, reason="forced 40-bit runs always take the parse")
def test_automatic_producer_goes_by_the_number_of_documents():
    import mumemto_amd
    e = mumemto_amd.Engine(0)
    for n_docs, expected in ((3, "direct"), (4, "direct"), (5, "pfp"), (9, "pfp")):
        e.set_docs(synth.pangenome(n_docs, 8000, 0.01, seed=n_docs))
        e.run()
        assert producer_is(e, expected)
    e.close()


def test_the_expansion_fixture_expands(engine, request):
    """the "expand" engines really sort representatives and run the emitter behind them, the others do not"""
    engine.set_docs(synth.pangenome(7, 9000, 0.01, seed=3))
    engine.run()
    param = request.node.callspec.params["engine"]
    assert engine.producer_expanded() == param.startswith("expand")
    if param.startswith("expand"):
        assert engine.producer_used() == "guided" and engine.stream_stats()["windows"] >= 3


def check_stream(engine, docs, revcomp):
    text, doc_start = O.build_text(docs, revcomp)
    sa, lcp, bwt = O.build_stream(text)
    assert engine.text_length() == len(text)
    assert np.array_equal(engine.text(), text)
    assert np.array_equal(engine.sa().astype(np.int64), sa[1:])
    assert np.array_equal(engine.lcp().astype(np.int64), lcp[1:])
    assert np.array_equal(engine.bwt(), bwt[1:])
    return sa, lcp, bwt, doc_start


CASES = {
    "snp": dict(n_haps=5, length=20000, divergence=0.01, seed=1),
    "indel_inv": dict(n_haps=6, length=30000, divergence=0.01, seed=2, indel_rate=0.002, inversion=(2, 4000, 8000)),
    "tandem": dict(n_haps=4, length=15000, divergence=0.005, seed=3, tandem=(1, 3000, 3400, 5)),
    "n_run_lower": dict(n_haps=4, length=12000, divergence=0.02, seed=4, n_run=(0, 2000, 2600), lowercase_frac=0.1),
    "identical": dict(n_haps=3, length=5000, divergence=0.0, seed=5),
}
MODES = {
    "mum": dict(num_distinct=0, max_doc_freq=1, max_total_freq=0),
    "partial": dict(num_distinct=-1, max_doc_freq=1, max_total_freq=0),
    "mem_f2": dict(num_distinct=0, max_doc_freq=2, max_total_freq=0),
    "mem_k-1_f3": dict(num_distinct=-1, max_doc_freq=3, max_total_freq=0),
    "mem_unlimited": dict(num_distinct=2, max_doc_freq=0, max_total_freq=0),
    "mem_capped": dict(num_distinct=2, max_doc_freq=0, max_total_freq=7),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("revcomp", [True, False])
def test_stream_columns(engine, case, revcomp):
    docs = synth.pangenome(**CASES[case])
    engine.set_docs(docs)
    engine.run(use_revcomp=revcomp)
    check_stream(engine, docs, revcomp)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("revcomp", [True, False])
def test_output_bytes(engine, case, mode, revcomp):
    docs = synth.pangenome(**CASES[case])
    m = dict(MODES[mode])
    if m["num_distinct"] < 0:
        m["num_distinct"] = len(docs) + m["num_distinct"]
    engine.set_docs(docs)
    engine.run(min_match_len=20, use_revcomp=revcomp, **m)
    want = O.run(docs, min_len=20, revcomp=revcomp, **m)
    assert engine.output_text() == want.text()
    if m["max_doc_freq"] == 1:
        L, off, st = engine.rows_mum()
        wl, wo, ws = want.mum_rows()
        assert np.array_equal(L, wl) and np.array_equal(off, wo) and np.array_equal(st, ws)
        assert engine.output_bumbl() == want.bumbl()
    else:
        L, occ, off, ids, st = engine.rows_mem()
        wl, wocc, wo, wd, ws = want.mem_rows()
        assert np.array_equal(L, wl) and np.array_equal(occ.astype(np.int64), wocc)
        assert np.array_equal(off, wo) and np.array_equal(ids.astype(np.int64), wd) and np.array_equal(st, ws)


@pytest.mark.parametrize("case", ["snp", "indel_inv", "tandem"])
def test_merge_thresholds(engine, case):
    docs = synth.pangenome(**CASES[case])
    engine.set_docs(docs)
    engine.run(merge_metadata=True)
    want = O.run(docs, merge=True)
    assert engine.output_text() == want.text()
    assert np.array_equal(engine.thresholds(), want.thresh())


def test_known_answer_vectors_through_c_abi():
    import mumemto_amd
    from mumsfile import parse_mums
    spec = json.load(open(os.path.join(HERE, "golden", "toy_vectors.json")))
    for v in spec["vectors"]:
        docs = [[r.encode() for r in d] for d in v["docs"]]
        want = O.run(docs, min_len=v["min_len"], revcomp=v["revcomp"], max_doc_freq=v["max_doc_freq"])
        assert want.text() == v["expect"].encode()
        if v["max_doc_freq"] == 1:
            got = mumemto_amd.mumemto_mum(docs, v["min_len"], v["revcomp"])
            wl, wo, ws = want.mum_rows()
            assert np.array_equal(got["lengths"], wl)
            assert np.array_equal(got["offsets"].reshape(wo.shape), wo)
            assert np.array_equal(got["strands"].reshape(ws.shape), ws)
            if len(wl):
                pl, po, ps = parse_mums(v["expect"].encode())
                assert np.array_equal(got["lengths"], pl) and np.array_equal(got["offsets"], po)
        else:
            got = mumemto_amd.mumemto_mem(docs, v["min_len"], v["revcomp"], max_doc_freq=v["max_doc_freq"])
            wl, wocc, wo, wd, ws = want.mem_rows()
            assert [m["length"] for m in got["mems"]] == list(wl)
            flat = np.concatenate([m["offsets"] for m in got["mems"]]) if got["mems"] else np.zeros(0, np.int64)
            assert np.array_equal(flat, wo)
        assert got["record_lengths"] == [[len(r) for r in d] for d in docs]


def test_c_abi_error_codes_and_edge_inputs():
    import ctypes as C
    import mumemto_amd
    L = mumemto_amd.load_library()
    out = C.c_void_p()
    assert L.mumemto_mum(None, 0, 20, 1, 0, 0, None) == 1           # out_result == NULL
    assert L.mumemto_mum(None, 2, 20, 1, 0, 0, C.byref(out)) == 2   # docs == NULL && num_docs != 0
    assert out.value is None
    assert L.mumemto_mum(None, 0, 20, 1, 0, 0, C.byref(out)) == 0   # empty -> empty result
    assert L.num_mums(out) == 0 and L.num_docs(out) == 0
    L.mum_free(out)
    with pytest.raises(mumemto_amd.MumemtoError):                    # f <= 1 in MEM mode -> rc 3
        mumemto_amd.mumemto_mem([[b"ACGT"], [b"ACGT"]], max_doc_freq=1)
    assert b"must be > 1" in L.mumemto_last_error()
    assert L.num_mums(None) == 0 and L.mum_at(None, 3).length == 0
    # multi-record docs, an empty record and a doc without any base
    docs = [[b"ACGTTGCATTGACCAGTAGGCTA", b"", b"GGATCCATTGACCAGTAGGCTAAC"], [b"TTGACCAGTAGGCTAAGG"], [b""]]
    got = mumemto_amd.mumemto_mum(docs, 5, True, num_distinct=2)
    want = O.run(docs, min_len=5, revcomp=True, num_distinct=2)
    wl, wo, ws = want.mum_rows()
    assert np.array_equal(got["lengths"], wl) and np.array_equal(got["offsets"].reshape(wo.shape), wo)
    assert got["record_lengths"] == [[23, 0, 24], [18], [0]]


def test_many_documents_counter_path(engine):
    # > 64 documents: the verifier uses LDS counters instead of the one-wave bitmap
    docs = synth.pangenome(70, 1500, 0.01, seed=8)
    for m in (dict(num_distinct=0, max_doc_freq=1), dict(num_distinct=60, max_doc_freq=1),
              dict(num_distinct=65, max_doc_freq=2)):
        engine.set_docs(docs)
        engine.run(**m)
        assert engine.output_text() == O.run(docs, **m).text()


@pytest.mark.parametrize("n_docs", [65, 94, 128])
def test_block_window_scan_for_65_to_128_documents(engine, n_docs):
    """Strict multi-MUMs with a window of 64 .. 127 entries (the 94-haplotype shape) take the form of k_scan whose window
    minima come from block prefix / suffix minima and whose BWT test is a running maximum (kernels.hip, VH)."""
    for seed, length, div, indel, at_least in ((21, 6000, 0.0003, 0.0, 20), (22, 3000, 0.002, 0.0002, 0)):
        docs = synth.pangenome(n_docs, length, div, seed=seed + n_docs, indel_rate=indel)
        for revcomp in (True, False):
            engine.set_docs(docs)
            engine.run(min_match_len=12, use_revcomp=revcomp)
            want = O.run(docs, min_len=12, revcomp=revcomp).text()
            assert engine.output_text() == want and want.count(b"\n") >= at_least
            # the same scanned in ranges of 8192 suffix-array positions (tiles that start inside a range, left extensions)
            os.environ["MMT_SCAN_RANGE"] = "8192"
            os.environ["MMT_FORCE_WIDE"] = "1"
            try:
                engine.run(min_match_len=12, use_revcomp=revcomp)
            finally:
                del os.environ["MMT_SCAN_RANGE"], os.environ["MMT_FORCE_WIDE"]
            assert engine.output_text() == want


def test_thousands_of_documents(engine):
    # per-wave LDS counters of 4 bytes per document: 6000 documents need 96 KiB of dynamic LDS per workgroup
    rng = np.random.default_rng(3)
    core = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=28))
    docs = []
    for d in range(6000):
        flank = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(3, 9))))
        docs.append([flank + (core if d % 7 else core[:20] + b"A") + flank[::-1]])
    for m in (dict(num_distinct=4000, max_doc_freq=2, max_total_freq=0), dict(num_distinct=3000, max_doc_freq=1)):
        engine.set_docs(docs)
        engine.run(min_match_len=12, **m)
        assert engine.output_text() == O.run(docs, min_len=12, **m).text()
    # strict multi-MUMs of 1,800 documents that do share the core: the exact-window form of the wide scan
    # (k_scan_wide: the window of num_distinct - 1 entries no longer fits k_scan's LDS tile)
    sharing = [d for i, d in enumerate(docs) if i % 7][:1800]
    engine.set_docs(sharing)
    engine.run(min_match_len=12, num_distinct=0, max_doc_freq=1)
    expected = O.run(sharing, min_len=12, num_distinct=0, max_doc_freq=1).text()
    assert engine.output_text() == expected and expected.count(b"\n") >= 1


def test_wide_window_scan_on_random_collections():
    """MMT_SCAN_WIDE_AT=1 sends every scan with more than two documents through the path for > 1000 documents (sliding
    minima from block prefix / suffix minima in HBM, k_scan_wide): the differential test over random collections and
    parameters, both producers."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_run.py"), "4200", "4", "40"],
                       env=dict(os.environ, MMT_SCAN_WIDE_AT="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_tiny_and_degenerate_texts(engine):
    for docs in ([[b"A"], [b"A"]], [[b""], [b""]], [[b"ACGT" * 30], [b"ACGT" * 30]], [[b"A" * 200], [b"A" * 150]],
                 [[b"N" * 50 + b"ACGTGGA" * 5], [b"ACGTGGA" * 5 + b"N" * 40]]):
        for revcomp in (True, False):
            for m in (dict(max_doc_freq=1), dict(max_doc_freq=3), dict(max_doc_freq=0, num_distinct=2)):
                engine.set_docs(docs)
                engine.run(min_match_len=4, use_revcomp=revcomp, **m)
                check_stream(engine, docs, revcomp)
                assert engine.output_text() == O.run(docs, min_len=4, revcomp=revcomp, **m).text(), (docs, revcomp, m)


def test_anchor_merge_against_reference_binary(engine):
    import glob
    from mumsfile import parse_mums
    G = os.path.join(HERE, "golden", "anchor_merge")
    for case in sorted(os.listdir(G)):
        parts = []
        for p in sorted(glob.glob(os.path.join(G, case, "p*.mums"))):
            L, off, st = parse_mums(open(p, "rb").read())
            parts.append((L, off, st, np.fromfile(p[:-5] + ".athresh", np.uint16)))
        got = engine.anchor_merge(parts)
        assert got["text"] == open(os.path.join(G, case, "merged.mums"), "rb").read()
        assert got["thresh"].tobytes() == open(os.path.join(G, case, "merged.athresh"), "rb").read()


def test_partition_merge_equals_direct_run(engine):
    # partitions sharing doc 0 -> fold -> re-sort == direct run, byte for byte
    docs = synth.pangenome(7, 20000, 0.01, seed=21, indel_rate=0.001, inversion=(4, 3000, 5000))
    groups = [[0, 1, 2], [0, 3, 4], [0, 5, 6]]
    parts = []
    for g in groups:
        engine.set_docs([docs[i] for i in g])
        engine.run(merge_metadata=True)
        L, off, st = engine.rows_mum()
        parts.append((L, off, st, engine.thresholds()[: len(docs[0][0]) + 1].copy()))
    merged = engine.anchor_merge(parts, sort_like_direct=True)   # engine's last run holds the anchor ranks
    direct = O.run(docs, merge=True)
    assert merged["text"] == direct.text()
    assert np.array_equal(merged["thresh"], direct.thresh()[: len(docs[0][0]) + 1])


@pytest.mark.parametrize("min_len", [10, 50])
def test_partition_merge_with_another_min_len_equals_direct_run(engine, min_len):
    # the reference's anchor_merge hard-codes 20 (merge_candidates.cpp:141); partitions made with another -l fold
    # to the direct run of that -l only when the fold is told the same length
    docs = synth.pangenome(7, 20000, 0.004, seed=22, indel_rate=0.0005)
    groups = [[0, 1, 2, 3], [0, 4, 5, 6]]
    parts = []
    for g in groups:
        engine.set_docs([docs[i] for i in g])
        engine.run(min_match_len=min_len, merge_metadata=True)
        L, off, st = engine.rows_mum()
        parts.append((L, off, st, engine.thresholds()[: len(docs[0][0]) + 1].copy()))
    merged = engine.anchor_merge(parts, sort_like_direct=True, min_len=min_len)
    direct = O.run(docs, min_len=min_len, merge=True)
    assert merged["text"] == direct.text() and merged["text"].count(b"\n") > 20


def test_full_size_properties(engine):
    # larger than the oracle comfortably checks in CI time: size-independent properties
    docs = synth.pangenome(8, 400000, 0.005, seed=33)
    engine.set_docs(docs)
    engine.run()
    sa = engine.sa().astype(np.int64)
    n = engine.text_length()
    assert n == 8 * 2 * (400000 + 1)
    assert np.array_equal(np.sort(sa), np.arange(n))                 # a permutation
    text = engine.text()
    lcp = engine.lcp().astype(np.int64)
    idx = np.random.default_rng(0).integers(1, n, size=20000)
    for j in idx[:2000]:                                             # sortedness + exact lcp on a sample
        a, b, l = sa[j - 1], sa[j], lcp[j]
        assert np.array_equal(text[a:a + l], text[b:b + l])
        ca = text[a + l] if a + l < n else -1
        cb = text[b + l] if b + l < n else -1
        assert ca < cb
    L, off, st = engine.rows_mum()
    assert len(L) > 100
    for r in np.random.default_rng(1).integers(0, len(L), size=200):  # every reported occurrence is a real match
        seqs = []
        for d in range(8):
            s = docs[d][0][off[r, d]: off[r, d] + L[r]]
            if not st[r, d]:
                s = s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
            seqs.append(s)
        assert len(set(seqs)) == 1 and len(seqs[0]) == L[r]
    # idempotence: same input, same bytes
    first = engine.output_text()
    engine.run()
    assert engine.output_text() == first


def test_two_fingerprint_phrase_grouping_path():
    """The PFP producer orders phrases by one 64-bit fingerprint and repeats the grouping with two when different
    phrases share it (verified, ~1e-5 per run); MMT_PFP_TWO_FINGERPRINTS forces that path."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MMT_PFP_TWO_FINGERPRINTS="1")
    r = subprocess.run([sys.executable, os.path.join(here, "scan_shape_check.py"), "5", "40000"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "scan shapes ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("env", [{"MMT_PFP_NO_PACK": "1"}, {"MMT_LONG_CAP": "3"}, {"MMT_PFP_NO_BWT_CODE": "1"}, {}])
def test_long_matches_and_rare_construction_paths(env):
    """Exact document copies and a long tandem repeat give irreducible LCP values far beyond the 192 characters one lane
    compares (k_long_lcp, and k_huge_lcp beyond 64 KB); MMT_LONG_CAP forces the overflow-and-rerun of the long-match list, MMT_PFP_NO_PACK the
    dictionary records without the packed previous byte (>= 2^24 distinct phrases in production), MMT_PFP_NO_BWT_CODE the
    oversized emitter groups whose BWT byte is read from the text instead of riding in the sort key (> 16 kinds of bytes)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "scan_shape_check.py"), "5", "60000", "dups"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "scan shapes ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("producer", ["pfp", "direct"])
@pytest.mark.parametrize("giant", ["3000", None, "3000 old route", "run buckets"])
def test_letter_runs_make_giant_sort_ranges(producer, giant):
    """Runs of N / homopolymers put tens of thousands of suffixes into one bucket of a doubling round (and one group of
    the PFP emitter); ranges beyond MMT_GIANT_RANGE elements are sorted device-wide instead of by one workgroup of the
    segmented sort (prims.hip sort_ranges)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MUMEMTO_PRODUCER=producer)
    if giant == "run buckets":        # every bucket of a long run of one symbol ordered by the end of its run (sorter.cpp refine_runs)
        env["MMT_RUN_BUCKET"] = "2"
    elif giant:
        env["MMT_GIANT_RANGE"] = giant.split()[0]
        if "old route" in giant:      # a sort per giant range + a segmented sort for the rest, instead of all ranges as one sort
            env["MMT_RANGES_AS_ONE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(here, "scan_shape_check.py"), "5", "30000", "runs"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "scan shapes ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("producer", ["pfp", "direct"])
@pytest.mark.parametrize("env", [{"MMT_SORT_FUSED": "0"}, {"MMT_ROUND_CAP": "1024", "MMT_GIANT_RANGE": "3000"},
                                 {"MMT_SORT_ONE_STREAM": "1", "MMT_SORT_ALL_RANKS": "1"}, {"MMT_ROUND_CAP": "1536"},
                                 {"MMT_ROUND_CAP": "1024", "MMT_BIG_CAP": "1"},
                                 {"MMT_ROUND_CAP": "1024", "MMT_RANGES_AS_ONE": "0"}, {"MMT_SORT_FUSED": "0", "MMT_RANGES_AS_ONE": "0"},
                                 {"MMT_ROUND_CAP": "1024", "MMT_NO_RUN_REFINE": "1"}])
def test_doubling_round_paths(producer, env):
    """A doubling round of the suffix sorter is one pass over the active list (k_round_fused + the scatter of the changed
    ranks on a second stream), with the ranges beyond an LDS tile finished around a segmented sort (k_big_*).  The
    round of separate kernels (MMT_SORT_FUSED=0), the smaller tiles (more ranges take the long path), the scatter of
    every rank on the one stream, a list of long ranges that overflows (MMT_BIG_CAP=1: the round of separate kernels
    takes over after the fused pass has run): all must give the bytes of the oracle, on letter runs (one bucket of tens of thousands
    of suffixes through many rounds) and on exact copies."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "scan_shape_check.py"), "5", "30000", "runs,dups"],
                       env=dict(os.environ, MUMEMTO_PRODUCER=producer, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("scan shapes ok") == 2, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("variant,bpc", [(0, 1), (0, 16), (1, 1), (2, 2)])
def test_scan_kernel_shapes_and_double_buffering(variant, bpc):
    """k_scan's workgroup shape and grid size are tuning knobs (MMT_SCAN_VARIANT / MMT_SCAN_BPC, read once per
    process).  With one workgroup per CU every workgroup walks several tiles of a 1.4 M-character text, so the
    LDS-DMA double buffering, the interior fast path and the boundary tiles all run; outputs must not change."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MMT_SCAN_VARIANT=str(variant), MMT_SCAN_BPC=str(bpc))
    r = subprocess.run([sys.executable, os.path.join(here, "scan_shape_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "scan shapes ok" in r.stdout

