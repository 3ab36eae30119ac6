"""Sequence content that i.i.d. bases lack (SURVEY.md 8(d) realism knobs, src/ref_builder.cpp:29-38 keeps N / IUPAC):
satellite arrays (period 171), microsatellites (period 2 - 6: no trigger of the parse falls inside a periodic run,
include/newscan.hpp:265-325 -> giant phrases), assembly gaps (runs of N), indels and inversions between the haplotypes
-- through every producer against the CPU oracle, and at the size of BASELINE configs[2] by properties and by the
agreement of the two independent producers."""
import os

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth
from conftest import producer_is, PACKED_TEXT

pytestmark = pytest.mark.gpu


def _docs(haps, length, div, seed, **kw):
    return [[s.tobytes()] for _, s in synth.haplotypes_realistic(haps, length, div, seed, **kw)]


@pytest.mark.parametrize("producer", ["pfp", "guided", "direct"])
def test_realistic_content_equals_the_oracle(producer):
    import mumemto_amd
    docs = _docs(6, 400_000, 0.002, 21, indel_rate=2e-4, inversion_every=3)
    eng = mumemto_amd.Engine(0)
    if producer == "guided":
        os.environ["MMT_GUIDED_BATCH"] = "150000"
    try:
        eng.set_producer(producer, 10, 30) if producer != "direct" else eng.set_producer("direct")
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(), dict(num_distinct=5, max_doc_freq=3), dict(num_distinct=2, max_doc_freq=0, max_total_freq=30),
                   dict(merge_metadata=True)):
            eng.set_docs(docs)
            eng.run(**kw)
            okw = dict(kw)
            merge = okw.pop("merge_metadata", False)
            want = O.run(docs, merge=merge, **okw)
            assert eng.output_text() == want.text(), (producer, kw)
            if merge:
                assert np.array_equal(eng.thresholds(), want.thresh())
        assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
        assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(eng.bwt(), bwt[1:])
    finally:
        os.environ.pop("MMT_GUIDED_BATCH", None)
        eng.close()


@pytest.mark.parametrize("env", [dict(), dict(MMT_GIANT_PROMOTE="0"), dict(MMT_GUIDED_NO_EARLY_GIANT="1"),
                                 dict(MMT_GUIDED_NO_TAIL="1", MMT_GUIDED_NO_SMALL_LCP="1", MMT_GUIDED_NO_ROUND_LCP="1"),
                                 dict(MMT_GUIDED_NO_DENSE="1", MMT_PACKED_TEXT="1"), dict(MMT_PACKED_TEXT="1")])
def test_giant_phrases_through_expansion_equal_the_oracle(env):
    """Round 6's routes through the giant dictionary, at a size the oracle sorts: with the giant depth at two key lengths (42
    characters) every microsatellite, satellite stretch and gap of the realistic haplotypes lies in giant phrases, so that groups take
    their order from the giant dictionary EARLY (k_giant_probe: all members in its phrases), the dictionary holds the NEIGHBOURS of giant
    occurrences (build_giant, MMT_GIANT_PROMOTE), the keys of a giant round are entries whose LCPs k_round_heads reads off the
    dictionary's LCP array, k_resolve_small leaves LCPs, the tail of the rounds is finished by comparison -- each with its old path
    beside it (the variables), on bytes and on packed text (plain tiles read as 2-bit words).  Stream, rows and thresholds."""
    import mumemto_amd
    docs = _docs(7, 300_000, 0.002, 33, indel_rate=2e-4, inversion_every=3)
    eng = mumemto_amd.Engine(0)
    os.environ.update(MMT_GUIDED_BATCH="120000", MMT_GIANT_DEPTH="2", **env)
    try:
        eng.set_producer("expand", 10, 30)
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(merge_metadata=True), dict(num_distinct=6, max_doc_freq=3, max_total_freq=21)):
            eng.set_docs(docs)
            eng.run(**kw)
            assert eng.producer_used() == "guided" and eng.producer_expanded()
            okw = dict(kw)
            merge = okw.pop("merge_metadata", False)
            want = O.run(docs, merge=merge, **okw)
            assert eng.output_text() == want.text(), (env, kw)
            if merge:
                assert np.array_equal(eng.thresholds(), want.thresh())
        assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
        assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(eng.bwt(), bwt[1:])
    finally:
        for k in ["MMT_GUIDED_BATCH", "MMT_GIANT_DEPTH"] + list(env):
            os.environ.pop(k, None)
        eng.close()


@pytest.mark.parametrize("haps,length", [(12, 30_000), (40, 9_000), (130, 3_000)])
def test_bucket_wise_producer_on_many_copies_equals_the_oracle(haps, length):
    """Groups of 9 .. 128 copies of a position are finished by one wave per group (guided_kernels.hip k_resolve_medium),
    larger ones by the refinement rounds: both against the oracle, columns included."""
    import mumemto_amd
    docs = synth.pangenome(haps, length, 0.004, seed=haps, indel_rate=0.001, tandem=(1, 500, 700, 4))
    eng = mumemto_amd.Engine(0)
    os.environ["MMT_GUIDED_BATCH"] = "60000"
    try:
        eng.set_producer("guided", 10, 30)
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(), dict(num_distinct=haps - 1, max_doc_freq=3)):
            eng.set_docs(docs)
            eng.run(**kw)
            assert eng.producer_used() == "guided"
            assert eng.output_text() == O.run(docs, **kw).text(), (haps, kw)
        assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
        assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(eng.bwt(), bwt[1:])
    finally:
        os.environ.pop("MMT_GUIDED_BATCH", None)
        eng.close()


@pytest.mark.parametrize("depth,haps,length,small", [(1, 6, 120_000, True), (2, 12, 40_000, True), (24, 6, 300_000, True),
                                                      (1, 40, 9_000, False), (5, 130, 3_000, True), (4, 20, 30_000, False)])
def test_giant_phrases_of_the_bucket_wise_producer_equal_the_oracle(depth, haps, length, small):
    """Phrases longer than MMT_GIANT_DEPTH first-key lengths (24 by default: runs of N, microsatellites) are sorted once as
    a dictionary of their own and every comparison that is still undecided there continues on its ranks and its LCP array
    (guided.cpp build_giant, guided_kernels.hip cmp_rest).  A depth of one or two key lengths makes most phrases of a
    small case giant, so that the small-group, medium-group, round and LCP paths all go through the structure; with and
    without the small-group resolvers."""
    import mumemto_amd
    docs = _docs(haps, length, 0.003, 100 + haps, indel_rate=3e-4, inversion_every=4)
    eng = mumemto_amd.Engine(0)
    os.environ["MMT_GUIDED_BATCH"] = "90000"
    os.environ["MMT_GIANT_DEPTH"] = str(depth)
    if not small:
        os.environ["MMT_GUIDED_NO_SMALL"] = "1"
    try:
        eng.set_producer("guided", 10, 30)
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(), dict(num_distinct=max(2, haps - 2), max_doc_freq=2)):
            eng.set_docs(docs)
            eng.run(**kw)
            assert eng.producer_used() == "guided"
            assert eng.output_text() == O.run(docs, **kw).text(), (depth, haps, kw)
        assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
        assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(eng.bwt(), bwt[1:])
    finally:
        for k in ("MMT_GUIDED_BATCH", "MMT_GIANT_DEPTH", "MMT_GUIDED_NO_SMALL"):
            os.environ.pop(k, None)
        eng.close()


def _flat(haps, length, div, seed):
    # (a haplotype per worker process: indels and inversions are minutes of numpy at whole-genome size)
    return synth.collection_realistic(haps, length, div, seed, procs=2 if length > 1_000_000_000 else 8)


def test_realistic_collection_of_c3_size():
    """94 x ~64 Mbp with satellite arrays (1.5 and 2.5 Mbp of a 171-base monomer), microsatellites, gaps of up to 1 Mbp,
    indels and inversions -- 12.0 G text characters as one suffix array through the parse proper (wide positions, oversized
    groups of the emitter, giant phrases in the dictionary), through the bucket-wise producer (giant phrases as a dictionary of
    their own) and as anchor partitions + merge: the same bytes three ways, rows checked against the definition."""
    import bigchecks
    import mumemto_amd
    from test_gpu_fullsize import _partitioned, _same_up_to_the_stream_end_quirk
    bases, lens = _flat(94, 64_000_000, 0.001, 3)
    eng = mumemto_amd.Engine(0)
    try:
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1
        assert eng.is_wide() and producer_is(eng, "pfp") and (PACKED_TEXT or eng.pfp_counts()["oversized_groups"] > 1000)
        single = eng.output_text()
        assert single.count(b"\n") > 100_000
        bigchecks.check_mum_rows(eng, bases, lens)
        eng.set_producer("guided")
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1 and eng.producer_used() == "guided"
        assert eng.output_text() == single
        eng.set_producer("auto")
        parts, part = _partitioned(eng, bases, lens, 0.36)
        assert parts >= 3 and _same_up_to_the_stream_end_quirk(single, part, parts)
        # partial multi-MEMs with the parameters of BASELINE configs[4]
        assert eng.run_partitioned(None, flat=(bases, lens), num_distinct=93, max_doc_freq=3) == 1
        bigchecks.check_mem_rows(eng, bases, lens, min_docs=93, max_doc_freq=3)
    finally:
        eng.close()


def test_realistic_anchor_next_to_one_whole_genome_haplotype():
    """{anchor, one haplotype} of ~3.05 Gbp each with the same structures (gaps of 2.4 / 9.5 / 48 Mbp, satellite arrays of 70
    and 119 Mbp): the unit of the anchor-merge workflow for BASELINE configs[3]; the automatic producer is the bucket-wise one."""
    import bigchecks
    import mumemto_amd
    bases, lens = _flat(2, 3_050_000_000, 0.001, 11)
    eng = mumemto_amd.Engine(0)
    try:
        eng.keep_columns(True)
        assert eng.run_partitioned(None, flat=(bases, lens)) == 1
        assert eng.is_wide() and eng.producer_used() == "guided"
        bigchecks.check_stream(eng, bases, lens, light=True)
        bigchecks.check_mum_rows(eng, bases, lens)
    finally:
        eng.close()


@pytest.mark.parametrize("seed,bucket", [(1, "2"), (2, "2"), (3, "64"), (4, None)])
def test_long_runs_of_one_symbol_in_the_dictionary_equal_the_oracle(seed, bucket):
    """Runs of 21 or more equal symbols -- assembly gaps, homopolymers -- put every suffix inside them into ONE bucket of the
    dictionary's first sort; the sorter orders that bucket once by (class of the symbol behind the run, length of the run that is
    left, the symbols behind it) instead of doubling through it (sorter.cpp refine_runs, kernels.hip k_run_keys).  Here: runs of
    every symbol, of lengths around the key length and far beyond it, at the ends of documents, next to each other, copied
    between haplotypes with substitutions inside and behind them, and broken by insertions -- stream and rows against the
    oracle through the parse proper, with the bucket threshold lowered so that small buckets take the path too."""
    import mumemto_amd
    rng = np.random.default_rng(900 + seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = acgt[rng.integers(0, 4, size=60_000)].copy()
    for _ in range(60):
        c = b"ACGTN"[int(rng.integers(0, 5))]
        k = int(rng.choice([20, 21, 22, 23, 40, 64, 100, 333, 1500]))
        a = int(rng.integers(0, len(anc) - k))
        anc[a:a + k] = c
    anc[:400] = ord("N"); anc[-300:] = ord("A")             # runs at both ends of every document
    docs = []
    for h in range(5):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=120)
        s[pos] = acgt[rng.integers(0, 4, size=len(pos))]     # substitutions: also inside runs (they break them)
        cut = int(rng.integers(0, 50))
        docs.append([s[cut:].tobytes()])
    if bucket is not None:
        os.environ["MMT_RUN_BUCKET"] = bucket
    eng = mumemto_amd.Engine(0)
    try:
        eng.set_producer("pfp", 10, 30)
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(), dict(num_distinct=4, max_doc_freq=2)):
            eng.set_docs(docs)
            eng.run(**kw)
            assert producer_is(eng, "pfp")
            assert eng.output_text() == O.run(docs, **kw).text(), (seed, kw)
        assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
        assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(eng.bwt(), bwt[1:])
        if bucket is not None:
            assert eng.pfp_counts()["run_refined"] > 0
    finally:
        os.environ.pop("MMT_RUN_BUCKET", None)
        eng.close()


def _gap_docs(seed):
    """haplotypes with assembly gaps and homopolymers the way whole genomes carry them: the same gap in every haplotype with
    private lengths (indels break gaps), a gap behind which a T follows (T > N: the other class of the closed form), gaps at both
    ends of a document, a run of A (its reverse complement: a run of T), substitutions next to and inside the runs"""
    rng = np.random.default_rng(4000 + seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = acgt[rng.integers(0, 4, size=24_000)].copy()
    docs = []
    for h in range(6):
        s = anc.copy()
        pos = rng.integers(0, len(s), size=80)
        s[pos] = acgt[rng.integers(0, 4, size=len(pos))]
        parts = [s[:3000], np.full(6000 + 37 * h, ord("N"), np.uint8), s[3000:9000],
                 np.full(3000 - 11 * h, ord("N"), np.uint8), np.frombuffer(b"T", np.uint8), s[9000:16000],
                 np.full(4000 + (h % 3), ord("A"), np.uint8), s[16000:]]
        if h == 2:
            parts.insert(0, np.full(2500, ord("N"), np.uint8))          # a gap that begins the document ...
        if h == 4:
            parts.append(np.full(1700, ord("N"), np.uint8))             # ... and one that ends it (followed by '$' < N)
        if h == 5:
            parts[1] = np.concatenate([np.full(3500, ord("N"), np.uint8), acgt[rng.integers(0, 4, size=7)], np.full(2500, ord("N"), np.uint8)])
        docs.append([np.concatenate(parts).tobytes()])
    return docs


@pytest.mark.parametrize("seed,limit,wp", [(1, "3000", (6, 16)), (3, "20000", (6, 16)), (4, "3000", (11, 7))])
def test_bins_of_one_repeated_symbol_are_produced_in_slices(seed, limit, wp):
    """The suffixes that begin with N^4 -- every position inside every assembly gap, on both strands -- are ONE bin of the
    bucket-wise producer whatever the number of leading characters: 1.5 G suffixes in a rank's share of 13 whole genomes with 60 Mbp
    of gaps each, more than a batch holds.  Such a bin is produced in slices by what is left of the run (guided_kernels.hip RunSlice:
    c^r X sorts by (X0 < c, r) in closed form).  MMT_GUIDED_SLICE forces slices of a few thousand suffixes here: stream, rows and
    thresholds against the oracle in the capped modes; the uncapped mode (an interval may be as long as its bin) keeps the bin whole.
    The parse (11, 7) is the one the closed form does not hold for: the hash of N^11 is divisible by 7, so a phrase ends at every
    position of a gap and the occurrences of a representative lie all over the bin -- its N bin stays whole (guided.cpp run_triggers;
    the automatic parameters avoid such moduli)."""
    import mumemto_amd
    docs = _gap_docs(seed)
    eng = mumemto_amd.Engine(0)
    os.environ["MMT_GUIDED_BATCH"] = "9000"
    os.environ["MMT_GUIDED_SLICE"] = limit
    try:
        eng.set_producer("expand", *wp)
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        for kw in (dict(merge_metadata=True), dict(num_distinct=5, max_doc_freq=3, max_total_freq=18), dict(use_revcomp=False)):
            eng.set_docs(docs)
            eng.run(**kw)
            assert eng.producer_used() == "guided" and eng.producer_expanded()
            st = eng.producer_stats()
            assert st["run_slices"] >= (6 if limit != "20000" and wp == (6, 16) else 2), st
            assert st["staged"] and st["batches"] >= (10 if wp == (6, 16) else 5), st
            okw = dict(kw)
            merge = okw.pop("merge_metadata", False)
            revcomp = okw.pop("use_revcomp", True)
            want = O.run(docs, merge=merge, revcomp=revcomp, **okw)
            assert eng.output_text() == want.text(), (seed, limit, wp, kw)
            if merge:
                assert np.array_equal(eng.thresholds(), want.thresh())
            if revcomp:
                assert np.array_equal(eng.sa().astype(np.int64), sa[1:])
                assert np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
                assert np.array_equal(eng.bwt(), bwt[1:])
        # uncapped: no slices, same bytes as the oracle
        eng.set_docs(docs)
        eng.run(num_distinct=2, max_doc_freq=0)
        assert eng.producer_stats()["run_slices"] == 0
        assert eng.output_text() == O.run(docs, num_distinct=2, max_doc_freq=0).text()
    finally:
        os.environ.pop("MMT_GUIDED_BATCH", None)
        os.environ.pop("MMT_GUIDED_SLICE", None)
        eng.close()
