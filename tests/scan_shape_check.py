"""GPU box helper (run in a subprocess with MMT_SCAN_VARIANT / MMT_SCAN_BPC set): a text long enough that
every workgroup of k_scan processes several tiles -- i.e. takes the LDS-DMA double-buffered path -- checked
against the oracle in the strict (exact-window) mode, with merge metadata, and in partial / MEM modes."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import pyoracle as O                       # noqa: E402
import mumemto_amd                         # noqa: E402
from mumemto_amd import synth              # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
length = int(sys.argv[2]) if len(sys.argv) > 2 else 120_000
# (several shapes in one process, "runs,dups": the HIP runtime, the library and numpy come up once)
shapes = sys.argv[3].split(",") if len(sys.argv) > 3 else [""]
eng = mumemto_amd.Engine(0)
for shape in shapes:
    docs = synth.pangenome(n_docs, length, 0.01, seed=31)
    if shape == "dups":
        # exact copies and a long tandem repeat: irreducible LCP values beyond 100,000 characters -- past the 64 KB one wave
        # compares in k_long_lcp, into k_huge_lcp
        docs[1] = [docs[0][0]]
        docs[2] = [docs[0][0][: length // 2] + docs[0][0][: length // 2]]
    if shape == "runs":
        # runs of one letter (assembly gaps, homopolymers): tens of thousands of suffixes share every prefix a doubling round
        # has seen, i.e. one bucket far beyond an LDS tile; with MMT_GIANT_RANGE small such a range takes the device-wide sort
        docs[1] = [docs[1][0][:1000] + b"N" * 20000 + docs[1][0][1000:]]
        docs[2] = [docs[2][0][:7000] + b"N" * 9000 + docs[2][0][7000:]]
        docs[3] = [b"A" * 15000 + docs[3][0][:5000]]
    eng.set_docs(docs)
    cases = [
        dict(min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, merge_metadata=False),
        dict(min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, merge_metadata=True),
        dict(min_match_len=12, num_distinct=n_docs - 1, max_doc_freq=1, max_total_freq=0, merge_metadata=False),
        dict(min_match_len=25, num_distinct=2, max_doc_freq=3, max_total_freq=0, merge_metadata=False),
        dict(min_match_len=30, num_distinct=2, max_doc_freq=0, max_total_freq=7, merge_metadata=False),
    ]
    for c in cases:
        eng.run(use_revcomp=True, **c)
        ref = O.run(docs, min_len=c["min_match_len"], num_distinct=c["num_distinct"], max_doc_freq=c["max_doc_freq"],
                    max_total_freq=c["max_total_freq"], revcomp=True, merge=c["merge_metadata"])
        got = eng.output_text()
        assert got == ref.text(), ("output differs", shape, c, len(got), len(ref.text()))
        assert got.count(b"\n") > 0 or shape in ("dups", "runs"), c
        if shape in ("dups", "runs"):      # the stream itself, column by column
            import numpy as np
            text, _ = O.build_text(docs, True)
            sa, lcp, bwt = O.build_stream(text)
            assert np.array_equal(eng.sa().astype(np.int64), sa[1:]) and np.array_equal(eng.lcp().astype(np.int64), lcp[1:])
            assert np.array_equal(eng.bwt(), bwt[1:]) and int(lcp.max()) > (2 * length if shape == "dups" else 8000)
        if c["merge_metadata"]:
            import numpy as np
            assert np.array_equal(eng.thresholds(), ref.thresh())
    print("scan shapes ok: %s text of %d characters, %d cases" % (shape, eng.text_length(), len(cases)))
