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
