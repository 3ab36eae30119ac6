"""GPU: the text packed to two bits per character (mumemto_amd/csrc/textref.hpp) -- what lets one device hold a collection of
hundreds of G characters (BASELINE configs[4]: every rank holds all 573 G; profiles/round4_c5_rank_share_250G_packed.log).
MMT_PACKED_TEXT=1 forces the layout at any size: A C G T in two bits, everything else ('$', N, IUPAC codes) as sorted
exception runs; the parse and the bucket-wise producer read it through one accessor, and every result must be the byte
layout's.  MMT_INPUT_DEFERRED=1 also forces the route whose raw bases never sit on the device as a whole (packed document by
document through a staging buffer).  The whole parity suite runs with MMT_PACKED_TEXT=1 as well (profiles/README.md)."""
import os

import numpy as np
import pytest

import mumemto_amd.binding
import pyoracle as O
from mumemto_amd import synth

pytestmark = pytest.mark.gpu


def _awkward_docs():
    rng = np.random.default_rng(12)
    anc = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=30000).astype(np.uint8)
    docs = []
    for d in range(6):
        s = anc.copy()
        for p in rng.integers(0, len(s), size=60):
            s[p] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
        if d % 2 == 0:
            a = int(rng.integers(100, 20000))
            s[a:a + int(rng.integers(1, 5000))] = ord("N")          # runs of N: exception runs over several 4096-blocks
        for p in rng.integers(0, len(s), size=12):
            s[p] = rng.choice(np.frombuffer(b"RYKMSWBDHVN", np.uint8))  # single IUPAC codes
        if d == 3:
            s[:40] = ord("N"); s[-33:] = ord("N")                     # runs that touch both ends of a document
        rec = s.tobytes()
        if d == 4:
            rec = rec.lower()
        docs.append([rec[:7000], rec[7000:]] if d == 5 else [rec])
    return docs


class packed_env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("revcomp", [True, False])
def test_packed_text_spells_the_text_and_gives_the_same_rows(revcomp):
    import mumemto_amd
    docs = _awkward_docs()
    text, _ = O.build_text(docs, revcomp)
    eng = mumemto_amd.Engine(0)
    try:
        for kw in (dict(), dict(num_distinct=5, max_doc_freq=3, max_total_freq=18), dict(merge_metadata=True)):
            eng.set_docs(docs)
            eng.run(use_revcomp=revcomp, **kw)
            plain = eng.output_text()
            th = eng.thresholds().copy() if kw.get("merge_metadata") else None
            okw = {k: v for k, v in kw.items() if k != "merge_metadata"}
            want = O.run(docs, revcomp=revcomp, merge=bool(kw.get("merge_metadata")), **okw)
            assert plain == want.text()
            with packed_env(MMT_PACKED_TEXT=1):
                eng.set_docs(docs)
                eng.run(use_revcomp=revcomp, **kw)
                assert eng.producer_used() == "guided"
                assert np.array_equal(eng.text(), text), "the packed text does not spell T"
                assert eng.output_text() == plain
                if th is not None:
                    assert np.array_equal(eng.thresholds(), th)
                # the raw bases never on the device as a whole: document by document through a staging buffer
                with packed_env(MMT_INPUT_DEFERRED=1):
                    flat = np.frombuffer(b"".join(b"".join(d) for d in docs), np.uint8)
                    lens = np.array([sum(len(r) for r in d) for d in docs], np.uint64)
                    assert eng.run_partitioned(None, flat=(flat, lens), use_revcomp=revcomp,
                                               merge_metadata=bool(kw.get("merge_metadata")), **okw) == 1
                    assert np.array_equal(eng.text(), text)
                    assert eng.output_text() == plain
            # the deferred input with the byte layout (uploaded as a whole when the run begins)
            with packed_env(MMT_INPUT_DEFERRED=1, MMT_PACKED_TEXT=0):
                assert eng.run_partitioned(None, flat=(flat, lens), use_revcomp=revcomp,
                                           merge_metadata=bool(kw.get("merge_metadata")), **okw) == 1
                assert eng.output_text() == plain
    finally:
        eng.close()


def test_shards_of_a_packed_text_concatenate_to_the_single_run():
    import mumemto_amd
    docs = synth.pangenome(7, 40000, 0.01, seed=31, indel_rate=0.0005, inversion=(3, 3000, 9000))
    want = O.run(docs, num_distinct=6, max_doc_freq=3, max_total_freq=21).text()
    eng = mumemto_amd.Engine(0)
    try:
        with packed_env(MMT_PACKED_TEXT=1, MMT_GUIDED_BATCH=5000):
            got = b""
            for r in range(3):
                eng.set_docs(docs)
                eng.set_scan_shard(r, 3)
                eng.run(num_distinct=6, max_doc_freq=3, max_total_freq=21)
                assert eng.stream_stats()["entries"] == eng.sort_pieces()[r][1]
                got += eng.output_text()
            eng.set_scan_shard(0, 1)
        assert got == want and want.count(b"\n") > 20
    finally:
        eng.close()


def test_bytes_the_parse_reserves_keep_the_byte_layout():
    import mumemto_amd
    docs = [[b"ACGT\x01\x02ACGTTGCA" * 5], [b"ACGTTGCA\x01ACGT" * 4]]
    eng = mumemto_amd.Engine(0)
    try:
        with packed_env(MMT_PACKED_TEXT=1):
            eng.set_docs(docs)
            eng.run(min_match_len=4, num_distinct=2, max_doc_freq=3)
            assert eng.producer_used() == "direct"
            assert eng.output_text() == O.run(docs, min_len=4, num_distinct=2, max_doc_freq=3).text()
    finally:
        eng.close()


@pytest.mark.parametrize("producer", ["pfp", "guided"])
def test_rows_written_window_by_window_and_dropped(producer, tmp_path):
    """What a run over a text that fills the device does with its rows (Engine::sink_flush with `sink_discard_`): every window's
    accepted rows are formatted, sent to the file and FORGOTTEN -- in the multi-MEM modes too (a rank of BASELINE configs[4]
    accepts rows with ~94 occurrences each: tens of GB of suffix-array entries, offsets and text that would have to stay in
    HBM to the end).  The files are the oracle's; the engine answers for the row count only."""
    import mumemto_amd
    docs = synth.pangenome(7, 40000, 0.01, seed=33, indel_rate=0.0005, inversion=(3, 3000, 9000))
    eng = mumemto_amd.Engine(0)
    try:
        eng.set_producer(producer)
        for name, kw in (("mems", dict(num_distinct=6, max_doc_freq=3, max_total_freq=21)), ("mums", dict()),
                         ("mems", dict(num_distinct=2, max_doc_freq=0, max_total_freq=30))):
            want = O.run(docs, **kw).text()
            out = str(tmp_path / ("out." + name))
            with packed_env(MMT_SINK_DISCARD=1, MMT_SCAN_RANGE=8192, MMT_GUIDED_BATCH=6000):
                eng.set_text_sink(out)
                eng.set_docs(docs)
                eng.run(**kw)
                eng.set_text_sink(None)
            assert open(out, "rb").read() == want and want.count(b"\n") > 20
            assert eng.L.mmt_num_rows(eng.h) == want.count(b"\n")
            assert eng.stream_stats()["windows"] >= 4 and not os.path.exists(out + ".tmp")
            # the rows left with their windows: the accessors say so (they used to hand out null arrays)
            with pytest.raises(mumemto_amd.binding.MumemtoError, match="left the device window by window"):
                eng.rows_mem() if name == "mems" else eng.rows_mum()
            # without the sink the same engine keeps its rows as before
            eng.set_docs(docs)
            eng.run(**kw)
            assert eng.output_text() == want
    finally:
        os.environ.pop("MMT_GUIDED_NO_RANK", None)
        eng.set_producer("auto")
        eng.close()


@pytest.mark.parametrize("packed", [1, 0])
def test_documents_supplied_one_at_a_time(packed):
    """mmt_engine_run_supplied: the engine asks for the documents in order and packs (or uploads) each from its own page-locked
    buffer -- the host never holds the collection (BASELINE configs[4]: 287 GB of bases; the reference streams its FASTA files
    through the parser, include/newscan.hpp:265-325).  Same bytes as the oracle's run over the resident documents; a supplier
    that fails stops the run with its own exception."""
    import mumemto_amd
    docs = _awkward_docs()
    flat = [b"".join(d).upper() for d in docs]
    lens = np.array([len(f) for f in flat], np.uint64)
    asked = []

    def supplier(d, dst):
        asked.append(d)
        assert dst.dtype == np.uint8 and len(dst) == len(flat[d])
        dst[:] = np.frombuffer(flat[d], np.uint8)

    eng = mumemto_amd.Engine(0)
    try:
        with packed_env(MMT_PACKED_TEXT=packed):
            for kw in (dict(), dict(num_distinct=5, max_doc_freq=3)):
                asked.clear()
                eng.run_supplied(lens, supplier, **kw)
                assert asked == list(range(len(docs)))
                assert eng.output_text() == O.run(docs, **kw).text()
                assert bytes(eng.text()) == bytes(O.build_text(docs, True)[0])

            def broken(d, dst):
                if d == 3:
                    raise KeyError("no such document")
                supplier(d, dst)
            with pytest.raises(KeyError):
                eng.run_supplied(lens, broken)
            # the engine is usable afterwards
            eng.run_supplied(lens, supplier)
            assert eng.output_text() == O.run(docs).text()
            # two documents (the direct producer: the raw bases are uploaded as a whole, through the same supplier)
            for k in (2,):
                asked.clear()
                eng.run_supplied(lens[:k], supplier, num_distinct=k, min_match_len=12)
                assert asked == list(range(k))
                assert eng.output_text() == O.run(docs[:k], num_distinct=k, min_len=12).text()
    finally:
        eng.close()


@pytest.mark.parametrize("packed,copies,length", [(1, 6, 50000), (0, 6, 50000), (1, 18, 14000), (0, 18, 14000)])
def test_long_phrases_and_shared_variants_in_the_groups_of_copies(packed, copies, length):
    """The bucket-wise producer on what a text of hundreds of G characters looks like to it (modulus 157: phrases of ~170
    characters) with variants that are SHARED by descent: 24 or 72 haplotypes (groups of up to 32 / up to 128 copies: two
    instantiations of the kernel) in four clades, clade variants + private ones.  The
    copies of a locus form one group; its members are compared once with a reference member chosen by majority
    (k_resolve_medium), 64 characters a round trip; members that share a variant differ from the reference at the same place
    with the same character and are compared further in the text.  Bytes of the oracle's, both text layouts."""
    import mumemto_amd
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    anc = rng.choice(acgt, size=length).astype(np.uint8)

    def mutate(s, k):
        s = s.copy()
        pos = rng.integers(0, len(s), size=k)
        s[pos] = acgt[(np.searchsorted(acgt, s[pos]) + rng.integers(1, 4, size=k)) & 3]
        return s
    docs = []
    for clade in range(4):
        base = mutate(anc, length // 400)
        for _ in range(copies):
            docs.append([mutate(base, length // 1200).tobytes()])
    eng = mumemto_amd.Engine(0)
    try:
        with packed_env(MMT_PACKED_TEXT=packed):
            for w, p, no_rank in ((14, 157, 1), (10, 60, 0)):       # (MMT_GUIDED_NO_RANK: the records of a text beyond ~2^33 characters)
                os.environ["MMT_GUIDED_NO_RANK"] = "1" if no_rank else "0"
                if not no_rank:
                    os.environ.pop("MMT_GUIDED_NO_RANK")
                eng.set_producer("guided", w, p)
                for kw in (dict(), dict(num_distinct=len(docs) - 1, max_doc_freq=3), dict(num_distinct=len(docs) // 2, max_doc_freq=2)):
                    eng.set_docs(docs)
                    eng.run(**kw)
                    assert eng.producer_used() == "guided"
                    want = O.run(docs, **kw).text()
                    assert eng.output_text() == want and want.count(b"\n") > 50
    finally:
        os.environ.pop("MMT_GUIDED_NO_RANK", None)
        eng.set_producer("auto")
        eng.close()


@pytest.mark.parametrize("stage,packed", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_passes_over_the_text_for_the_batches(stage, packed):
    """The bucket-wise producer finds the suffixes of a batch by a pass over the whole text.  MMT_GUIDED_STAGE=1 (automatic from
    text characters x batches = 2 x 10^13 on: a rank of BASELINE configs[4]): one pass writes the suffixes of the next batches
    to a staging list, every batch takes its own from it, and the pass also counts the next pass's suffixes per tile;
    MMT_GUIDED_STAGE=0: one pass per batch, whose fill counts the next batch's tiles.  Bytes of the oracle's either way."""
    import mumemto_amd
    docs = synth.pangenome(9, 40000, 0.01, seed=41, indel_rate=0.0005, inversion=(4, 3000, 9000))
    eng = mumemto_amd.Engine(0)
    try:
        eng.set_producer("guided")
        with packed_env(MMT_GUIDED_BATCH=30000, MMT_GUIDED_STAGE=stage, MMT_PACKED_TEXT=packed, MMT_SCAN_RANGE=16384):
            for kw in (dict(), dict(num_distinct=8, max_doc_freq=3, max_total_freq=27)):
                eng.set_docs(docs)
                eng.run(**kw)
                assert eng.producer_used() == "guided"
                assert eng.output_text() == O.run(docs, **kw).text()
                assert eng.stream_stats()["windows"] >= 8
    finally:
        eng.set_producer("auto")
        eng.close()
