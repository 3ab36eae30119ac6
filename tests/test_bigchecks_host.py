"""CPU: the helpers the full-size GPU scripts trust (tests/bigchecks.py).  `SparseModel` must yield exactly what
`mumemto_amd.synth.haplotypes_sparse` yields -- the configs[4] script supplies its documents from it one at a time and checks
sampled rows against it -- and `LazyText` must read the same text from the model as from the resident bases."""
import numpy as np

import bigchecks
from mumemto_amd import synth


def test_sparse_model_is_the_generator_and_lazy_text_reads_it():
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
