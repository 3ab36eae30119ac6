"""GPU: thresholds are 32 bits wide inside the engine, the fold and the exchange; only PREFIX.athresh / .thresh and
the 16-bit accessors saturate at 65535 like the reference's column (include/mem_finder.hpp:299,328).

The case where the widths differ: the anchor A and document B share a 72,000-base stretch X; B also holds a second copy of
X's first 71,000 bases; document C holds X's first 66,000 bases.  Partition {A, B} finds the multi-MUM X (72,000) with a
threshold of 71,000 at its anchor position (the next-best match), partition {A, C} finds X[:66000].  The fold proposes
min(72000, 66000) = 66,000 there and accepts it iff it exceeds the merged threshold max(71000, .) -- it does not: the
66,000-base string occurs TWICE in B, and the direct run on {A, B, C} reports no such row.  With 16-bit thresholds the
71,000 reads 65,535 and the row is accepted unproven: that is what the reference's `anchor_merge` does (the oracle's fold
restates it, and shows it here), and what this engine's 32-bit columns avoid."""
import os
import subprocess
import sys

import numpy as np
import pytest

import pyoracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _docs():
    rng = np.random.default_rng(2024)
    r = lambda n: rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).astype(np.uint8).tobytes()
    X, Y, Z = r(72_000), r(700), r(300)               # Y, Z: ordinary multi-MUMs of all three documents
    A = r(3_000) + X + r(1_000) + Y + r(2_000) + Z + r(500)
    B = r(2_500) + Y + r(900) + X + r(2_000) + X[:71_000] + r(2_500) + Z + r(100)
    Cd = Z + r(3_500) + X[:66_000] + r(3_500) + Y + r(800)
    return [[A], [B], [Cd]], len(A)


def test_fold_with_32_bit_thresholds_equals_the_direct_run_where_16_bits_saturate():
    import mumemto_amd
    docs, L0 = _docs()
    direct = O.run(docs).text()
    eng = mumemto_amd.Engine(0)
    try:
        eng.set_docs(docs)
        eng.run()
        assert eng.output_text() == direct and direct.count(b"\n") >= 2
        parts16, parts32 = [], []
        for g in ([0, 2], [0, 1]):                    # (the last run = partition 0: the engine keeps its anchor ranks)
            eng.set_docs([docs[i] for i in g])
            eng.run(merge_metadata=True)
            l, o, s = (x.copy() for x in eng.rows_mum())
            t32, t16 = eng.thresholds32()[: L0 + 1].copy(), eng.thresholds()[: L0 + 1].copy()
            assert t32.dtype == np.uint32 and t16.dtype == np.uint16
            assert np.array_equal(np.minimum(t32, 65535).astype(np.uint16), t16)
            # the oracle's column is the reference's: 16 bits, saturated
            assert np.array_equal(t16, O.run([docs[i] for i in g], merge=True).thresh()[: L0 + 1])
            parts32.insert(0, (l, o, s, t32)); parts16.insert(0, (l, o, s, t16))
        assert int(parts32[0][3].max()) == 71_000 and int(parts16[0][3].max()) == 65_535
        m32 = eng.anchor_merge(parts32, sort_like_direct=True)
        m16 = eng.anchor_merge(parts16, sort_like_direct=True)
        assert m32["text"] == direct, "the fold at 32 bits is the direct run"
        extra = set(m16["text"].split(b"\n")) - set(direct.split(b"\n"))
        # (66,000 bases, or a few more when C's next bases happen to continue X)
        assert len(extra) == 1 and 66000 <= int(next(iter(extra)).split(b"\t")[0]) < 66020, "the 16-bit fold accepts the unproven 66,000-base row"
        # ... exactly as the reference's anchor_merge does (oracle restatement of src/merge_candidates.cpp)
        om = O.anchor_merge([(p[0], p[1], p[2], p[3]) for p in parts16])
        assert any(66000 <= int(x) < 66020 for x in om[0])
        # by coordinate ranges, and from device-resident 32-bit columns, the same
        assert eng.anchor_merge(parts32, sort_like_direct=True, slices=3)["text"] == direct
        # the 16-bit file form of the merged thresholds saturates, the rows do not depend on it
        assert m32["thresh"].dtype == np.uint16 and int(m32["thresh"].max()) <= 65535
    finally:
        eng.close()


def test_exchange_carries_32_bit_thresholds(tmp_path):
    """The same collection through `mumemto_exec --gpus 2` (ranks share GPU 0 over the transport double): the native exchange
    sends the 32-bit columns, so the merged PREFIX.mums is the direct run's; PREFIX.athresh is 16 bits and saturates."""
    from mumemto_amd import synth
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.dirname(lib)])
    docs, L0 = _docs()
    paths = []
    for i, d in enumerate(docs):
        p = str(tmp_path / ("d%d.fa" % i))
        synth.write_fasta(p, d)
        paths.append(p)
    env = dict(os.environ, MUMEMTO_RCCL_LIB=lib, MUMEMTO_SHARE_DEVICE="1")
    exe = os.path.join(ROOT, "mumemto_amd", "bin", "mumemto_exec")
    for fold in ("0", "1"):
        out = str(tmp_path / ("out" + fold))
        r = subprocess.run([exe, "--gpus", "2", "-n", "-o", out] + paths, capture_output=True, text=True, timeout=600,
                           env=dict(env, MUMEMTO_RANGE_FOLD=fold))
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(out + ".mums", "rb").read() == O.run(docs).text()
        th = np.fromfile(out + ".athresh", np.uint16)
        assert len(th) == L0 + 1 and int(th.max()) <= 65535
