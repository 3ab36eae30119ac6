"""GPU: the HIP path held DIRECTLY against the definition checker (tests/bruteforce.py) -- not by way of the oracle.

The oracle (oracle/mumemto_oracle.c) restates mem_finder.hpp's stack scan; tests/bruteforce.py applies the definitions of
SURVEY.md 8(a) A5 / A6 / A7 to every repeated substring and shares nothing with either.  tests/test_oracle_fuzz.py holds the
oracle against the checker on the CPU (10,000 cases); here the same seeded cases -- 100 per mode, 500 in all, the cases of that
file: reverse-complement palindromes at document ends, tandem duplications, runs of N, IUPAC codes, lower-case and two-record
documents, min_len 4 - 20 -- go through libmumemto's C ABI on the GPU and their PREFIX.mums / .mems bytes must equal the
checker's: two independent checkers bracket the GPU path itself.  In the strict mode every fourth case also compares the
thresholds of A7 with their definition.  The checker is slow (seconds per case): the expected bytes are computed on the
host's cores in parallel while the GPU runs."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from test_oracle_fuzz import MODES, make_case, _thresholds_by_definition

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("MMT_GPU_BRUTEFORCE_CASES", "100"))


def _expected(args):
    mode, seed = args
    import pyoracle as O                       # (only its text layout: build_text restates src/ref_builder.cpp:211-314)
    from bruteforce import bruteforce_lines
    docs, revcomp, min_len, nd, f, F = make_case(seed, mode)
    text, doc_start = O.build_text(docs, revcomp)
    want = bruteforce_lines(text, list(doc_start), min_len, nd, f, F, revcomp)
    th = _thresholds_by_definition(text, list(doc_start), min_len, len(docs)) if mode == "mum" and seed % 4 == 0 else None
    return mode, seed, want, th


@pytest.mark.parametrize("producer", ["auto", "expand"])
def test_hip_path_equals_the_definition_checker(producer):
    import mumemto_amd
    jobs = [(mode, seed) for mode in MODES for seed in range(CASES)]
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    eng = mumemto_amd.Engine(0)
    if producer == "expand":
        eng.set_producer("expand", 4, 11)
        os.environ["MMT_GUIDED_BATCH"] = "400"
    bad, rows = [], 0
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            for mode, seed, want, th in pool.imap_unordered(_expected, jobs, chunksize=4):
                docs, revcomp, min_len, nd, f, F = make_case(seed, mode)
                eng.set_docs(docs)
                eng.run(min_match_len=min_len, num_distinct=nd, max_doc_freq=f, max_total_freq=F, use_revcomp=revcomp,
                        merge_metadata=(mode == "mum"))
                rows += want.count(b"\n")
                if eng.output_text() != want:
                    bad.append((mode, seed, "rows"))
                elif th is not None:
                    got = eng.thresholds()
                    if not np.array_equal(got[: len(th)], th) or got[len(th):].any():
                        bad.append((mode, seed, "thresholds"))
    finally:
        os.environ.pop("MMT_GUIDED_BATCH", None)
        eng.close()
    assert not bad, "HIP path != definition checker (mode, seed, what): %s" % bad[:10]
    assert rows > len(jobs) // 2
