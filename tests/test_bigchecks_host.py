"""The size-independent checkers of tests/bigchecks.py, validated on the CPU against the oracle's own stream and rows
(so that a green full-size GPU test means something): they accept the oracle's output and reject corrupted copies."""
import numpy as np
import pytest

import bigchecks
import pyoracle as O
from mumemto_amd import synth


class _OracleAsEngine:
    """The accessors bigchecks uses, answered by the CPU oracle."""
    def __init__(self, docs, **kw):
        self.text_arr, self.doc_start = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(self.text_arr)
        self._sa, self._lcp, self._bwt = np.asarray(sa)[1:].astype(np.uint64), np.asarray(lcp)[1:].astype(np.uint32), np.asarray(bwt)[1:].astype(np.uint8)
        self.res = O.scan(sa, lcp, bwt, self.doc_start, **kw)

    def text_length(self): return len(self._sa)
    def sa(self): return self._sa
    def lcp(self): return self._lcp
    def bwt(self): return self._bwt
    def rows_mum(self): return self.res.mum_rows()
    def rows_mem(self): return self.res.mem_rows()
    def output_text(self): return self.res.text()


def _flat(docs):
    return np.frombuffer(b"".join(d[0] for d in docs), np.uint8), np.array([len(d[0]) for d in docs], np.uint64)


def test_checkers_accept_the_oracle_and_reject_corruption():
    docs = synth.pangenome(6, 4000, 0.01, seed=5, inversion=(2, 800, 1500))
    bases, lens = _flat(docs)
    eng = _OracleAsEngine(docs)
    bigchecks.check_stream(eng, bases, lens, samples=200)
    bigchecks.check_mum_rows(eng, bases, lens)
    bad = _OracleAsEngine(docs)
    bad._sa = bad._sa.copy(); bad._sa[100] = bad._sa[101]                 # not a permutation any more
    with pytest.raises(AssertionError):
        bigchecks.check_stream(bad, bases, lens, samples=50)
    bad = _OracleAsEngine(docs)
    bad._lcp = bad._lcp.copy(); bad._lcp[1:60] += 1                       # a wrong LCP among the sampled head entries
    with pytest.raises(AssertionError):
        bigchecks.check_stream(bad, bases, lens, samples=50)


def test_mem_row_checker_on_partial_matches():
    docs = synth.pangenome(5, 3000, 0.02, seed=3, inversion=(1, 500, 1200))
    bases, lens = _flat(docs)
    eng = _OracleAsEngine(docs, num_distinct=4, max_doc_freq=3, max_total_freq=15)
    bigchecks.check_mem_rows(eng, bases, lens, min_docs=4, max_doc_freq=3)


def test_sparse_model_is_the_generator_and_lazy_text_reads_it():
    """`SparseModel` must yield exactly what `mumemto_amd.synth.haplotypes_sparse` yields -- the configs[4] script supplies its
    documents from it one at a time and checks sampled rows against it -- and `LazyText` must read the same text from the
    model as from the resident bases."""
    L, which = 200000, [0, 5, 7]
    ref = {h: b.copy() for h, b in synth.haplotypes_sparse(94, L, 0.01, 4, which=which)}
    m = bigchecks.SparseModel(94, L, 0.01, 4, which=which)
    for d, h in enumerate(which):
        dst = np.empty(L, np.uint8)
        m.fill(d, dst)
        assert np.array_equal(dst, ref[h])
        assert np.array_equal(m.doc(d)[1000:150000], ref[h][1000:150000]) and m.doc(d)[777] == ref[h][777]
    lens = [L] * 3
    flat = np.concatenate([ref[h] for h in which])
    a, b = bigchecks.LazyText(flat, lens), bigchecks.LazyText(m, lens)
    assert a.n == b.n == 3 * 2 * (L + 1)
    for lo, hi in [(0, 50), (L - 10, L + 20), (2 * L, 2 * L + 10), (2 * L - 3, 2 * L + 11), (5 * L, 5 * L + 100),
                   (6 * L, 6 * L + 36), (a.n - 20, a.n + 8)]:
        assert np.array_equal(a[lo:hi], b[lo:hi]), (lo, hi)
    rng = np.random.default_rng(1)
    for i in rng.integers(0, a.n, size=200):
        assert a[int(i)] == b[int(i)]
