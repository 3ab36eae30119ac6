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
    return synth.collection_sparse(haps, length, div, seed)         # (filled by a few threads: the same haplotypes)


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
    bases, lens = synth.collection_sparse(haps, length, 0.001, 4)
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


def _kmers_of(seq, k, count, seed, prefixes=None):
    """`count` distinct k-mers spelled at random places of the ASCII sequence `seq` (optionally: beginning with one of `prefixes`)"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        p = int(rng.integers(0, len(seq) - k))
        km = bytes(seq[p:p + k])
        if km in out or (prefixes and not km.startswith(tuple(prefixes))):
            continue
        out.append(km)
    return out


def test_two_rank_shares_of_configs3_and_their_fold_at_the_anchors_length():
    """BASELINE configs[3] on eight GPUs gives rank r {anchor + 12 (11)} x 3.05 Gbp = 79.3 G text characters: ONE streamed run
    with merge metadata (rows in anchor coordinates, 32-bit thresholds over the anchor, suffix ranks of the anchor for the
    re-sort), as `tests/big_c4.py` runs all eight shares and folds them (profiles/round5_*_c4_full.log).  Here the shares of
    ranks 0 and 1 run at their size, and their rows + thresholds are folded in eight slices at L0 = 3.05 G and re-sorted into
    direct-run order (src/merge_candidates.cpp:106-157, what ranks fold after the all-to-all).

    PRECISION AND RECALL of both shares inside whole bins (bigchecks.check_bins_complete): for sixteen 12-mers of the anchor the GPU
    lists every position of the 79.3 G-character text that begins with one, the host sorts those suffixes from the generator's
    model and runs the oracle's scan over that piece of the stream -- the intervals it reports must be exactly the ones the run
    accepted there.  Beside it the sampled properties: rows are real, maximal, one-per-document matches in lexicographic order,
    every sampled row's threshold sits at its anchor position and is shorter than the row; sampled merged rows are matches
    in all 25 documents, maximal, in the order of a direct run."""
    import mumemto_amd
    from mumemto_amd import dist as mdist
    length = 3_050_000_000
    groups = mdist.partition_docs(94, 8)
    eng = mumemto_amd.Engine(0)
    parts = []
    # (the second share's haplotypes are made on a host thread while the first share runs on the device)
    from concurrent.futures import ThreadPoolExecutor
    maker = ThreadPoolExecutor(1)
    made = {0: maker.submit(synth.collection_sparse, 94, length, 0.001, 4, groups[0])}
    for share in (0, 1):
        mine = groups[share]
        haps = len(mine)
        bases, lens = made.pop(share).result()
        if share == 0:
            made[1] = maker.submit(synth.collection_sparse, 94, length, 0.001, 4, groups[1], 4)
        # (a strict multi-MUM begins at one anchor position in a hundred: sixteen bins of 12 characters -- ~4700 suffixes each --
        # hold a few dozen rows between them)
        kmers = _kmers_of(bases[:length], 12, 16, seed=21 + share)
        eng.set_row_tap(kmers, max_rows=1 << 14, max_occ=1 << 20)
        assert eng.run_partitioned(None, flat=(bases, lens), merge_metadata=True) == 1
        assert eng.is_wide() and eng.text_length() == 2 * haps * (length + 1) > 79e9 and not eng.columns_kept()
        assert eng.producer_used() == "guided" and eng.producer_expanded()
        st = eng.stream_stats()
        assert st["entries"] == eng.text_length() and st["windows"] >= 30
        assert eng.device_memory()["peak"] < 0.87 * 288 * 2**30           # (head-room: a quarter-million-row collection is not the only one)
        text = bigchecks.LazyText(bases, lens)
        bins, suffixes, rows = bigchecks.check_bins_complete(eng, text, text.n, text.doc_start, kmers)
        print("share %d: %d bins of 12 characters, %d suffixes sorted on the host, %d rows: the run's rows there are exactly the oracle's"
              % (share, bins, suffixes, rows))
        assert bins == 16 and suffixes >= 16 * 2 * haps and rows >= 5
        eng.set_row_tap([])                                    # (switching the tap off forgets what it holds)
        if share == 0:
            bigchecks.check_mum_rows(eng, bases, lens, use_text=False)
            # the exchange's messages at THIS size through the real RCCL, the rank as its own peer (dist.cpp dist_loopback): 30 M x 13
            # row tables and the 3.05 G thresholds (12.2 GB) as grouped ncclSend / ncclRecv in pieces of 2^29 bytes, the all-gather
            # of the meta words, a broadcast -- counts beyond 2^31, size_t arithmetic and stream ordering meet the library itself
            comm = mumemto_amd.Comm(eng, 0, 1, mumemto_amd.Comm.unique_id())
            try:
                lb = comm.loopback()
            finally:
                comm.close()
            print("loopback through RCCL: %.1f GB in %d pieces (largest %.2f GB) in %.2f s, %d elements differ"
                  % (lb["bytes"] / 1e9, lb["pieces"], lb["largest_piece_bytes"] / 1e9, lb["seconds"], lb["different"]))
            assert lb["different"] == 0 and lb["thresholds"] == length + 1 and lb["rows"] > 25_000_000
            # (pieces of half a gibibyte: of a piece beyond 1 GiB half the elements arrived different in RCCL 2.26 -- this very
            # assertion found it at 2^30 ELEMENTS a piece; tests/micro/rccl_sizes.py, profiles/round6_rccl_piece_sizes.log)
            assert lb["largest_piece_bytes"] == 1 << 29 and lb["pieces"] >= 30
        L, off, strands = eng.rows_mum()
        assert len(L) > 25_000_000
        th = eng.thresholds32()[: length + 1].copy()
        rng = np.random.default_rng(5)
        for r in rng.integers(0, len(L), size=2000):
            t = int(th[int(off[r, 0])])                    # the anchor is '+' in every kept row: text offset = anchor offset
            assert 0 < t < int(L[r]), (r, t, int(L[r]))
        assert int(np.count_nonzero(th)) >= len(L)
        parts.append((L.copy(), off.copy(), strands.copy(), th))
        del bases
    # ---- the fold of the two shares, in eight slices of the anchor, re-sorted by the anchor's suffix ranks ----
    eng.release_columns(keep_anchor_ranks=True)
    m = eng.anchor_merge(parts, sort_like_direct=True, want_rows=True, slices=8, want_text=False)
    n_rows, n_docs = int(m["n_rows"]), int(m["n_docs"])
    # (a fold step starts a row wherever either partition does: more and shorter rows than either side has)
    assert n_docs == 25 and max(len(parts[0][0]), len(parts[1][0])) < n_rows <= len(parts[0][0]) + len(parts[1][0])
    order = mdist.merged_column_order(groups[:2])
    model = bigchecks.SparseModel(94, length, 0.001, 4, which=order)
    mtext = bigchecks.LazyText(model, np.full(n_docs, length, np.uint64))
    ml, mo, ms = m["lengths"], m["offsets"], m["strands"]
    comp = np.arange(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    rng = np.random.default_rng(9)
    keys = []
    for r in np.sort(rng.integers(0, n_rows, size=150)):
        ln = int(ml[r])
        segs, lefts, rights = set(), set(), set()
        for d in range(n_docs):
            o = int(mo[r, d])
            doc = model.doc(d)
            w = np.full(ln + 2, 36, np.uint8)
            lo, hi = max(o - 1, 0), min(o + ln + 1, length)
            w[lo - (o - 1):hi - (o - 1)] = doc[lo:hi]
            if ms[r, d]:
                segs.add(w[1:-1].tobytes()); lefts.add(int(w[0])); rights.add(int(w[-1]))
            else:
                segs.add(comp[w[1:-1][::-1]].tobytes()); lefts.add(int(comp[w[-1]])); rights.add(int(comp[w[0]]))
        assert len(segs) == 1 and ln >= 20, ("merged row is not a match in every document", r)
        assert len(lefts) > 1 and len(rights) > 1, ("merged row is not maximal", r)
        keys.append(next(iter(segs)))
    assert keys == sorted(keys), "merged rows are not in the order of a direct run"
    del mtext
    eng.close()
    eng.L.mmt_pool_trim()          # (250 GB of mapped heap would leave the command-line tests of other files, which run as
    #                                processes of their own on the same GPU, with nothing)


def test_a_rank_share_of_configs4_at_its_size():
    """BASELINE configs[4] -- 94 whole-genome haplotypes, partial multi-MEMs `-k -1 -f 3` (the mem_finder.hpp path) on eight GPUs --
    as rank 3 of 8 sees it: EVERY rank holds all 573.4 G characters (the reference refuses to merge these modes from partitions:
    include/pfp_mum.hpp:178-183), packed to two bits each; the documents are supplied one at a time (94 x 3.05 Gbp do not fit the
    host as bytes either); the rank produces, scans and drops its share of the stream -- whole bins of leading characters,
    12.4 % of the suffixes --; its ~44 M rows (66 GB of PREFIX.mems) are formatted, copied out, digested and dropped (the sink is
    /dev/null: the box has no room for them).

    PRECISION AND RECALL inside whole bins: twelve 14-mers of the rank's share; every position of the 573.4 G-character text that
    begins with one (~2100 each: 94 haplotypes, two strands, chance hits) is listed by the GPU, spelled from the generator's
    model and sorted on the host, and the oracle's scan over that piece of the stream (-k -1 -f 3: num_distinct 93, at most 3
    per document, 282 in all) must report exactly the intervals the run accepted there."""
    import mumemto_amd
    import pyoracle as O
    haps, length = 94, 3_050_000_000
    avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
    if avail_gb < 40:
        pytest.skip("the generator's model needs ~25 GB of host memory")
    model = bigchecks.SparseModel(haps, length, 0.001, 4, which=list(range(haps)))
    lens = np.full(haps, length, np.uint64)
    # rank 3 of 8 takes the suffixes between 37.5 % and 50 % of the sorted order: those that begin with CG or CT (the edges of
    # the share move by the few '$' and nothing else: k-mers well inside are asked for, and checked against the share afterwards)
    anchor = model.doc(0)[0:4_000_000]
    kmers = _kmers_of(anchor, 14, 12, seed=33, prefixes=[b"CGC", b"CGG", b"CGT", b"CTA", b"CTC", b"CTG"])
    mumemto_amd.load_library().mmt_pool_trim()      # (a text of 143 GB wants the heap in one piece: what earlier tests left goes first)
    eng = mumemto_amd.Engine(0)
    eng.set_scan_shard(3, 8)
    eng.set_text_sink("/dev/null")
    eng.set_row_tap(kmers, max_rows=1 << 16, max_occ=1 << 23)
    nd, f, mf = O.cli_params(haps, k=-1, f=3)
    assert (nd, f, mf) == (93, 3, 282)
    assert eng.run_supplied(lens, lambda d, dst: model.fill(d, dst), num_distinct=nd, max_doc_freq=f, max_total_freq=mf) == 1
    eng.set_text_sink(None)
    n_text = 2 * haps * (length + 1)
    assert eng.is_wide() and eng.text_length() == n_text > 573e9 and eng.producer_used() == "guided"
    pieces = eng.sort_pieces()
    st = eng.stream_stats()
    assert st["entries"] == pieces[3][1] and abs(pieces[3][1] / n_text - 1 / 8) < 0.02
    mem = eng.device_memory()
    assert mem["peak"] < 0.93 * 288 * 2**30, mem
    rows = eng.L.mmt_num_rows(eng.h)
    written, digest = eng.text_sink_digest()
    print("rank 3 of 8 of configs[4]: %d suffixes of %d in %d windows, %d rows, %.1f GB of PREFIX.mems (digest %016x), peak HBM %.1f GB"
          % (st["entries"], n_text, st["windows"], rows, written / 1e9, digest, mem["peak"] / 2**30))
    assert 30_000_000 < rows < 60_000_000 and written > 40e9
    assert all(eng.kmer_in_share(km) for km in kmers)
    text = bigchecks.LazyText(model, lens)
    bins, suffixes, tapped = bigchecks.check_bins_complete(eng, text, text.n, text.doc_start, kmers, num_distinct=nd, max_doc_freq=f,
                                                           max_total_freq=mf)
    print("%d bins of 14 characters, %d suffixes sorted on the host, %d rows: the run's rows there are exactly the oracle's" % (bins, suffixes, tapped))
    assert bins == 12 and suffixes >= 12 * 2 * haps - 40 and tapped >= 6
    eng.set_row_tap([])
    eng.close()
    eng.L.mmt_pool_trim()
