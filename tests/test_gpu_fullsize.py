"""BASELINE configurations at their full size (SURVEY.md 8(d) stand-ins), on the GPU.

C1 (3 x 4.64 Mbp) is small enough for the CPU oracle: byte-for-byte.  C2 (16 x 12.1 Mbp), a collection just beyond
2^32 text characters and C3 (94 x 64 Mbp, 12.0 G characters) are checked through size-independent properties
(tests/bigchecks.py: the suffix array is a permutation, sampled neighbours are in suffix order with the reported LCP
and BWT byte, sampled rows are real, maximal, one-per-document matches in lexicographic order) and by comparing the
two independent routes to the same answer: one suffix array (40-bit positions beyond 2^32 characters) against anchor
partitions + merge.
"""
import os

import numpy as np
import pytest

import bigchecks
from mumemto_amd import synth

pytestmark = pytest.mark.gpu


def _collection(haps, length, div, seed):
    bases = np.empty(haps * length, np.uint8)
    for h, b in synth.haplotypes_sparse(haps, length, div, seed):
        bases[h * length:(h + 1) * length] = b
    return bases, np.full(haps, length, np.uint64)


def _partitioned(eng, bases, lens, frac):
    n_text = int(sum(2 * (int(l) + 1) for l in lens))
    os.environ["MMT_MAX_TEXT"] = str(int(n_text * frac))
    try:
        parts = eng.run_partitioned(None, flat=(bases, lens))
    finally:
        del os.environ["MMT_MAX_TEXT"]
    return parts, eng.output_text()


def _same_up_to_the_stream_end_quirk(single, part, parts):
    """The reference never closes the last interval of a stream (pfp_lcp_mum.hpp:223-230): a partition whose last
    interval is a MUM loses that row, so partitions + merge may miss at most one row per partition."""
    if single == part:
        return True
    a, b = set(single.split(b"\n")), set(part.split(b"\n"))
    return len(b - a) == 0 and len(a - b) <= parts


def test_c1_standin_at_full_size_equals_the_cpu_oracle():
    import mumemto_amd
    import pyoracle as O
    docs = synth.pangenome(3, 4_640_000, 0.01, seed=1)
    eng = mumemto_amd.Engine(0)
    eng.set_docs(docs)
    eng.run()
    assert eng.output_text() == O.run(docs).text()


def test_c2_standin_at_full_size():
    import mumemto_amd
    bases, lens = _collection(16, 12_100_000, 0.005, 2)
    eng = mumemto_amd.Engine(0)
    eng.keep_columns(True)            # (the stream is produced window by window: whole columns only when asked for)
    assert eng.run_partitioned(None, flat=(bases, lens)) == 1
    assert not eng.is_wide() and eng.columns_kept()
    single = eng.output_text()
    bigchecks.check_stream(eng, bases, lens)
    eng.keep_columns(-1)
    bigchecks.check_mum_rows(eng, bases, lens)
    parts, part = _partitioned(eng, bases, lens, 0.4)
    assert parts >= 3 and _same_up_to_the_stream_end_quirk(single, part, parts)
    # the same collection through the 40-bit code path, scanned in ranges
    os.environ["MMT_FORCE_WIDE"] = "1"
    os.environ["MMT_SCAN_RANGE"] = str(1 << 26)
    try:
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1
        assert eng.is_wide() and eng.scan_ranges() >= 5
        assert eng.output_text() == single
    finally:
        del os.environ["MMT_FORCE_WIDE"], os.environ["MMT_SCAN_RANGE"]


def test_text_beyond_2_to_the_32_as_one_suffix_array():
    import mumemto_amd
    bases, lens = _collection(36, 60_000_000, 0.002, 7)            # |T| = 4.32 G characters > 2^32
    eng = mumemto_amd.Engine(0)
    eng.keep_columns(True)
    assert eng.run_partitioned(None, flat=(bases, lens)) == 1
    assert eng.is_wide() and eng.text_length() > 2 ** 32 and eng.scan_ranges() > 1
    single = eng.output_text()
    bigchecks.check_stream(eng, bases, lens)
    eng.keep_columns(-1)
    bigchecks.check_mum_rows(eng, bases, lens)
    parts, part = _partitioned(eng, bases, lens, 0.4)              # partitions of < 2^32 characters: the 32-bit path
    assert parts >= 3 and not eng.is_wide()
    assert _same_up_to_the_stream_end_quirk(single, part, parts)
    merged_thresh = eng.merged_thresholds(int(lens[0]))
    # merge metadata through the 40-bit path (what a rank of a 2-GPU run of the C3 stand-in does): every structural
    # interval is verified, thresholds are recorded; the direct run's .athresh equals the merged one
    import tempfile
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        paths = []
        for h in range(len(lens)):
            p = os.path.join(d, "h%02d.fa" % h)
            synth.write_fasta_fast(p, bases[h * 60_000_000:(h + 1) * 60_000_000], name="h%02d" % h)
            paths.append(p)
        eng.run_files(paths, out_prefix=os.path.join(d, "out"), merge_metadata=True)
        assert eng.is_wide()
        assert open(os.path.join(d, "out.mums"), "rb").read() == single
    direct_thresh = eng.thresholds()[: int(lens[0]) + 1]
    differ = np.nonzero(direct_thresh != merged_thresh)[0]
    # The fold of the partitions' thresholds is the direct run's .athresh except where the match of ALL documents is
    # shorter than -l: the direct run records nothing there (such intervals are never produced, mem_finder.hpp:350-353),
    # the partitions -- whose matches at that anchor position are longer -- do, and the fold keeps their maximum
    # (merge_candidates.cpp:121-123).  Same rows either way; tests/thresh_probe.py shows the same 8 entries of 12.1 M
    # on the C2 stand-in through the 32-bit path, the ranged scan and the 40-bit path.
    print("thresholds: %d of %d entries non-zero, %d differ between the direct run and the fold of %d partitions"
          % (int((direct_thresh > 0).sum()), len(direct_thresh), len(differ), parts))
    assert len(differ) < 1e-5 * len(direct_thresh)
    assert np.all(direct_thresh[differ] == 0) and np.all(merged_thresh[differ] >= 20)


def test_anchor_next_to_one_whole_genome_haplotype():
    """The partition the anchor-merge workflow needs for BASELINE configs[3]: {anchor, one other haplotype} of 3.05 Gbp
    each -- 12.2 G text characters whose two strands share nothing, so the dictionary of the parse (6+ G characters) is
    beyond a 32-bit suffix array and the automatic producer sorts the text suffixes themselves (guided.cpp)."""
    import mumemto_amd
    bases, lens = _collection(2, 3_050_000_000, 0.001, 11)
    eng = mumemto_amd.Engine(0)
    eng.keep_columns(True)
    assert eng.run_partitioned(None, flat=(bases, lens)) == 1
    assert eng.is_wide() and eng.text_length() == 4 * (3_050_000_000 + 1) and eng.producer_used() == "guided"
    bigchecks.check_stream(eng, bases, lens, light=True)
    bigchecks.check_mum_rows(eng, bases, lens)


def test_guided_producer_equals_the_parse_proper_at_c2_size():
    import mumemto_amd
    bases, lens = _collection(16, 12_100_000, 0.005, 2)
    eng = mumemto_amd.Engine(0)
    out = {}
    for kind, env in (("pfp", {}), ("guided", {}), ("guided", {"MMT_FORCE_WIDE": "1", "MMT_GUIDED_BATCH": str(50_000_000)})):
        eng.set_producer(kind)
        os.environ.update(env)
        try:
            assert eng.run_partitioned(None, flat=(bases, lens)) == 1
        finally:
            for k in env:
                del os.environ[k]
        assert eng.producer_used() == kind and eng.is_wide() == bool(env)
        out[kind + str(len(env))] = eng.output_text()
    assert out["guided0"] == out["pfp0"] and out["guided2"] == out["pfp0"]


def test_c3_standin_at_full_size_one_suffix_array():
    import mumemto_amd
    haps = 94
    bases, lens = _collection(haps, 64_000_000, 0.001, 3)          # |T| = 12.03 G characters
    eng = mumemto_amd.Engine(0)
    assert eng.run_partitioned(None, flat=(bases, lens)) == 1      # strict multi-MUMs, one suffix array
    assert eng.is_wide() and eng.text_length() == 2 * haps * (64_000_000 + 1)
    assert eng.stream_stats()["windows"] >= 40 and not eng.columns_kept()
    single = eng.output_text()
    bigchecks.check_mum_rows(eng, bases, lens)
    # the bytes `bench.py` times (its default workload is this collection; it prints config.output_sha256_16 and compares
    # with the same fixture): the driver-timed output is tied to the output checked here
    import hashlib, json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_standin_output.json")))
    assert len(single) == fx["output_bytes"] and single.count(b"\n") == fx["output_rows"]
    assert hashlib.sha256(single).hexdigest()[:16] == fx["output_sha256_16"]
    parts, part = _partitioned(eng, bases, lens, 0.36)
    assert parts >= 3 and _same_up_to_the_stream_end_quirk(single, part, parts)
    # the stream through the other producer (whole bins of leading characters, sorted batch by batch: guided.cpp) and
    # through windows a quarter the size: three independent ways of cutting the same stream into pieces that are
    # produced, scanned and dropped -- byte for byte the same output
    eng.set_producer("guided")
    try:
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1 and eng.producer_used() == "guided"
        assert eng.output_text() == single
        assert eng.stream_stats()["windows"] >= 10
    finally:
        eng.set_producer("auto")
    os.environ["MMT_SCAN_RANGE"] = str(1 << 26)
    try:
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1 and eng.producer_used() == "pfp"
        assert eng.output_text() == single and eng.stream_stats()["windows"] >= 170
    finally:
        del os.environ["MMT_SCAN_RANGE"]
    # partial multi-MEMs, the parameters of BASELINE configs[4] (-k -1 -f 3): one suffix array, no partitions possible
    assert eng.run_partitioned(None, flat=(bases, lens), num_distinct=haps - 1, max_doc_freq=3) == 1
    assert eng.is_wide()
    bigchecks.check_mem_rows(eng, bases, lens, min_docs=haps - 1, max_doc_freq=3)


def test_thirty_g_characters_as_one_streamed_run():
    """94 x 160 Mbp: 30.08 G text characters.  A stored stream (suffix array 5 + BWT 1 + LCP 4 bytes per character next to
    the text and the tables of the parse: 13.6 B per character in round 2) would need 409 GB; produced, scanned and dropped
    piece by piece (the reference does not store it either: include/pfp_lcp_mum.hpp:197) the run peaks at about half the
    device.  The tables of this collection's dictionary do not fit next to the text, so the automatic choice is the
    bucket-wise producer -- since round 5 with expansion: the collection is redundant, so one representative per distinct
    phrase suffix is sorted (a few batches) and the emitter of the parse proper writes the windows (34.7 -> 5.8 s)."""
    import mumemto_amd
    haps, length = 94, 160_000_000
    bases = np.empty(haps * length, np.uint8)
    for h, b in synth.haplotypes_sparse(haps, length, 0.001, 4):
        bases[h * length:(h + 1) * length] = b
    lens = np.full(haps, length, np.uint64)
    eng = mumemto_amd.Engine(0)
    assert eng.run_partitioned(None, flat=(bases, lens)) == 1
    assert eng.is_wide() and eng.text_length() == 2 * haps * (length + 1) > 30e9 and not eng.columns_kept()
    st = eng.stream_stats()
    assert st["entries"] == eng.text_length() and st["windows"] >= 10
    assert eng.producer_used() == "guided" and eng.producer_expanded()      # (redundant collection: representatives + the emitter)
    # (the heap's high-water mark is per process, i.e. of every test before this one: profiles/round3_a_30G_characters_one_gpu.log
    # has it for this run alone, 138 GB)
    assert st["window_bytes"] < 30e9, st
    bigchecks.check_mum_rows(eng, bases, lens)


def test_a_rank_share_of_configs3_anchor_and_twelve_whole_genome_haplotypes():
    """BASELINE configs[3] on eight GPUs gives rank 0 {anchor + 12} x 3.05 Gbp = 79.3 G text characters: ONE streamed run
    with merge metadata (rows in anchor coordinates, u16 thresholds over the anchor, suffix ranks of the anchor for the
    re-sort), as `tests/big_c4.py` runs all eight shares and folds them (profiles/round4_c4_full.log: 41.8 M merged rows
    x 94 columns, 49.7 GB of PREFIX.mums).  Checked by properties: sampled rows are real, maximal, one-per-document matches
    in lexicographic order; every sampled row's threshold sits at its anchor position and is shorter than the row."""
    import mumemto_amd
    haps, length = 13, 3_050_000_000
    bases = np.empty(haps * length, np.uint8)
    for h, b in synth.haplotypes_sparse(94, length, 0.001, 4, which=list(range(haps))):
        bases[h * length:(h + 1) * length] = b
    lens = np.full(haps, length, np.uint64)
    eng = mumemto_amd.Engine(0)
    assert eng.run_partitioned(None, flat=(bases, lens), merge_metadata=True) == 1
    assert eng.is_wide() and eng.text_length() == 2 * haps * (length + 1) > 79e9 and not eng.columns_kept()
    st = eng.stream_stats()
    assert st["entries"] == eng.text_length() and st["windows"] >= 30
    assert eng.producer_used() == "guided" and eng.producer_expanded()
    assert eng.device_memory()["peak"] < 288 * 2**30
    bigchecks.check_mum_rows(eng, bases, lens, use_text=False)
    L, off, strands = eng.rows_mum()
    assert len(L) > 25_000_000
    th = eng.thresholds()
    assert len(th) == 2 * (length + 1)
    rng = np.random.default_rng(5)
    for r in rng.integers(0, len(L), size=2000):
        t = int(th[int(off[r, 0])])                    # the anchor is '+' in every kept row: text offset = anchor offset
        assert 0 < t < int(L[r]), (r, t, int(L[r]))
    assert int(np.count_nonzero(th[: length + 1])) >= len(L)
    eng.close()
    eng.L.mmt_pool_trim()          # (250 GB of mapped heap would leave the command-line tests of other files, which run as
    #                                processes of their own on the same GPU, with nothing)
