"""The row tap + tests/bigchecks.py::check_bins_complete -- the instrument the full-size tests use for precision AND recall --
held against the oracle where the oracle reaches: small collections, every producer, every mode.  For a few k-mers the GPU
keeps a copy of every accepted interval whose match begins with one of them and lists every text position that begins with
one; the host sorts those suffixes, runs the oracle's scan (mem_finder.hpp:161-170,304-355 restated) over that piece of the
stream and the two sets of intervals must be equal -- and the instrument must notice a row that is missing, one too many, and
one with an occurrence missing."""
import os

import numpy as np
import pytest

import bigchecks
import pyoracle as O
from mumemto_amd import synth

pytestmark = pytest.mark.gpu

MODES = {
    "strict": dict(),
    "partial": dict(num_distinct=5, max_doc_freq=3),
    "mem": dict(num_distinct=2, max_doc_freq=0, max_total_freq=30),
    "metadata": dict(merge_metadata=True),
}


def _kmers(text, k, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        p = int(rng.integers(0, len(text) - k))
        km = bytes(text[p:p + k])
        if b"$" not in km and km not in out:
            out.append(km)
    return out


@pytest.fixture(scope="module")
def collection():
    docs = synth.pangenome(7, 24000, 0.01, seed=5, inversion=(3, 5000, 9000))
    text, doc_start = O.build_text(docs, True)
    return docs, text, doc_start


def _kmers_with_rows(docs, text, k, kw):
    """k-mers that real rows begin with -- four rows as the anchor's forward strand spells them, four as its reverse strand does
    (the twin of every multi-MUM: write_mum drops it, mem_finder.hpp:372-391, the scan accepts it) -- and four random ones"""
    okw = {a: b for a, b in kw.items() if a != "merge_metadata"}
    res = O.run(docs, **okw)
    iv = res.intervals()
    sa, _lcp, _bwt = O.build_stream(text)
    picked = []
    for s, e, l, _j in iv[:: max(1, len(iv) // 8)][:8]:
        p = int(sa[int(s)])
        km = bytes(text[p:p + k])
        if b"$" not in km and len(km) == k and km not in picked:
            picked.append(km)
    for km in _kmers(text, k, 12, seed=11):
        if len(picked) >= 12:
            break
        if km not in picked:
            picked.append(km)
    return picked


@pytest.mark.parametrize("producer", ["pfp", "direct", "guided", "expand"])
@pytest.mark.parametrize("k", [7, 12])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_bins_are_complete_on_a_small_collection(collection, producer, mode, k):
    import mumemto_amd
    docs, text, doc_start = collection
    kw = MODES[mode]
    kmers = _kmers_with_rows(docs, text, k, kw)
    eng = mumemto_amd.Engine(0)
    os.environ["MMT_GUIDED_BATCH"] = "30000"
    os.environ["MMT_SCAN_RANGE"] = "65536"
    try:
        eng.set_producer(producer, 6, 16) if producer != "direct" else eng.set_producer("direct")
        eng.set_row_tap(kmers)
        eng.set_docs(docs)
        eng.run(**kw)
        assert eng.output_text() == O.run(docs, **{k: v for k, v in kw.items() if k != "merge_metadata"}).text()
        okw = {k: v for k, v in kw.items() if k != "merge_metadata"}
        bins, suffixes, rows = bigchecks.check_bins_complete(eng, text, len(text), doc_start, kmers, **okw)
        assert bins == len(kmers) and suffixes >= len(kmers) and rows >= 4
    finally:
        del os.environ["MMT_GUIDED_BATCH"], os.environ["MMT_SCAN_RANGE"]
        eng.set_row_tap([])
        eng.close()


def test_the_instrument_notices_what_is_wrong(collection):
    import mumemto_amd
    docs, text, doc_start = collection
    kmers = _kmers_with_rows(docs, text, 7, dict(num_distinct=2, max_doc_freq=0, max_total_freq=30))
    eng = mumemto_amd.Engine(0)
    try:
        eng.set_row_tap(kmers)
        eng.set_docs(docs)
        eng.run(num_distinct=2, max_doc_freq=0, max_total_freq=30)
        okw = dict(num_distinct=2, max_doc_freq=0, max_total_freq=30)
        assert bigchecks.check_bins_complete(eng, text, len(text), doc_start, kmers, **okw)[2] > 3
        real = eng.row_tap()
        length, start, sa = real

        def without_row(r):
            keep = [i for i in range(len(length)) if i != r]
            cnt = np.diff(start.astype(np.int64))
            new_start = np.concatenate([[0], np.cumsum(cnt[keep])]).astype(np.uint64)
            new_sa = np.concatenate([sa[int(start[i]):int(start[i + 1])] for i in keep]) if keep else sa[:0]
            return length[keep], new_start, new_sa

        for broken in (without_row(0),                                               # a row is missing (recall)
                       (np.concatenate([length, length[:1] + 1]), np.concatenate([start, start[-1:] + (start[1] - start[0])]),
                        np.concatenate([sa, sa[int(start[0]):int(start[1])]])),        # a row nobody expects (precision)
                       (length, np.concatenate([start[:1], start[1:] - 1]), sa[1:])):  # the first row lost an occurrence
            eng.row_tap = lambda broken=broken: broken
            with pytest.raises(AssertionError):
                bigchecks.check_bins_complete(eng, text, len(text), doc_start, kmers, **okw)
        del eng.row_tap
        # a tap that overflows says so instead of returning a part
        eng.set_row_tap(kmers, max_rows=1, max_occ=4)
        eng.run(num_distinct=2, max_doc_freq=0, max_total_freq=30)
        with pytest.raises(mumemto_amd.binding.MumemtoError, match="overflowed"):
            eng.row_tap()
    finally:
        eng.set_row_tap([])
        eng.close()
