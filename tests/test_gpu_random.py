"""GPU differential testing: many small seeded random collections x random scan
parameters x both SA producers (random PFP w/p), output bytes against the oracle.
Targets the corner cases of the window logic in k_scan (every num_distinct / cap
combination), of k_verify (per-document caps) and of the PFP emitter."""
import numpy as np
import pytest

import pyoracle as O
from conftest import producer_is

pytestmark = pytest.mark.gpu
ALPH = np.frombuffer(b"ACGT", np.uint8)


def random_collection(rng):
    n_docs = int(rng.integers(2, 10))
    base = ALPH[rng.integers(0, 4, size=int(rng.integers(30, 1500)))]
    docs = []
    for _ in range(n_docs):
        s = base.copy()
        style = rng.integers(0, 6)
        n_mut = int(rng.integers(0, max(2, len(s) // 40)))
        for _ in range(n_mut):
            s[int(rng.integers(0, len(s)))] = rng.choice(np.frombuffer(b"ACGTNRY", np.uint8))
        if style == 0:      # reverse complement of the whole document
            comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78, 82: 89, 89: 82}
            s = np.array([comp[int(c)] for c in s[::-1]], np.uint8)
        elif style == 1:    # tandem duplication
            a = int(rng.integers(0, len(s) - 5)); b = min(len(s), a + int(rng.integers(3, 60)))
            s = np.concatenate([s[:b]] + [s[a:b]] * int(rng.integers(1, 6)) + [s[b:]])
        elif style == 2:    # truncated / shifted copy
            a = int(rng.integers(0, len(s) // 2))
            s = s[a:]
        elif style == 3:    # homopolymer run
            a = int(rng.integers(0, len(s)))
            s = np.concatenate([s[:a], np.full(int(rng.integers(5, 80)), s[a - 1] if a else 65, np.uint8), s[a:]])
        elif style == 4:    # unrelated document
            s = ALPH[rng.integers(0, 4, size=int(rng.integers(10, 300)))]
        if rng.random() < 0.2:
            s = np.array(bytearray(s.tobytes().lower()), np.uint8)
        recs = [s.tobytes()]
        if rng.random() < 0.3 and len(s) > 4:   # multi-record document
            c = int(rng.integers(1, len(s)))
            recs = [s[:c].tobytes(), s[c:].tobytes()]
        docs.append(recs)
    return docs


def random_params(rng, n_docs):
    min_len = int(rng.integers(3, 26))
    kind = rng.integers(0, 5)
    if kind == 0:
        return dict(min_len=min_len, num_distinct=n_docs, max_doc_freq=1, max_total_freq=0)
    if kind == 1:
        return dict(min_len=min_len, num_distinct=int(rng.integers(2, n_docs + 1)), max_doc_freq=1, max_total_freq=0)
    if kind == 2:
        f = int(rng.integers(2, 5))
        return dict(min_len=min_len, num_distinct=int(rng.integers(2, n_docs + 1)), max_doc_freq=f, max_total_freq=0)
    if kind == 3:
        f = int(rng.integers(0, 4))
        return dict(min_len=min_len, num_distinct=int(rng.integers(2, n_docs + 1)), max_doc_freq=f if f != 1 else 2,
                    max_total_freq=int(rng.integers(2, 3 * n_docs)))
    return dict(min_len=min_len, num_distinct=2, max_doc_freq=0, max_total_freq=0)


@pytest.mark.parametrize("seed", range(8))
def test_random_collections(seed):
    import mumemto_amd
    rng = np.random.default_rng(1000 + seed)
    eng = mumemto_amd.Engine(0)
    try:
        for case in range(40):
            docs = random_collection(rng)
            p = random_params(rng, len(docs))
            revcomp = bool(rng.integers(0, 2))
            merge = p["max_doc_freq"] == 1 and p["num_distinct"] == len(docs) and bool(rng.integers(0, 2))
            want = O.run(docs, revcomp=revcomp, merge=merge, **p)
            for producer in ("direct", "pfp", "guided"):
                wp = (int(rng.integers(2, 12)), int(rng.choice([3, 5, 7, 11, 13, 20, 37, 100])))
                eng.set_producer(producer, *wp)
                eng.set_docs(docs)
                eng.run(min_match_len=p["min_len"], num_distinct=p["num_distinct"], max_doc_freq=p["max_doc_freq"],
                        max_total_freq=p["max_total_freq"], use_revcomp=revcomp, merge_metadata=merge)
                assert eng.output_text() == want.text(), (seed, case, producer, wp, p, revcomp, docs)
                if merge:
                    assert np.array_equal(eng.thresholds(), want.thresh()), (seed, case, producer)
    finally:
        eng.close()


@pytest.mark.parametrize("env", [{}, {"MMT_GIANT_RANGE": "1500", "MMT_SCAN_WIDE_AT": "2", "MMT_LONG_CAP": "5",
                                      "MMT_GUIDED_BATCH": "3000", "MMT_GUIDED_NO_RANK": "1"}])
def test_mid_size_collections_with_runs_arrays_and_copies(env):
    """fuzz_run.py adv: haplotypes of 4-30 kbp with runs of N / of one base up to 12 kbp, tandem arrays, exact copies and
    deletions, random parameters, all three producers against the oracle; the second run lowers the thresholds of the
    device-wide range sort, of the wide scan and of the long-match list so that those paths take every case, and deals
    the guided producer's suffixes into batches of 3000 whose records carry phrase lengths instead of parse ranks."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_run.py"), "777", "1", "14", "adv"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_palindromic_ends_and_terminator_touching_matches():
    # matches running through the middle '$' (palindromic document end) are kept on '+',
    # '-' occurrences touching the terminator are dropped (mem_finder.hpp:372-373)
    import mumemto_amd
    pal = b"ACGTTGCAAGCTTGCAACGT"                       # its own reverse complement
    docs = [[b"TTGACCAGGATCCATA" + pal], [b"GGTTGACCAGGATCCATA" + pal], [pal + b"CCATGGAATTC"]]
    for revcomp in (True, False):
        for kw in (dict(num_distinct=2, max_doc_freq=1), dict(num_distinct=2, max_doc_freq=3),
                   dict(num_distinct=3, max_doc_freq=1)):
            got = mumemto_amd.mumemto_mum(docs, 6, revcomp, num_distinct=kw["num_distinct"]) \
                if kw["max_doc_freq"] == 1 else None
            want = O.run(docs, min_len=6, revcomp=revcomp, **kw)
            if got is not None:
                wl, wo, ws = want.mum_rows()
                assert np.array_equal(got["lengths"], wl) and np.array_equal(got["offsets"].reshape(wo.shape), wo)
            eng = mumemto_amd.Engine(0)
            eng.set_docs(docs)
            eng.run(min_match_len=6, use_revcomp=revcomp, **kw)
            assert eng.output_text() == want.text()
            eng.close()


def test_iupac_and_arbitrary_bytes():
    import mumemto_amd
    docs = [[b"ACGTRYKMSWBDHVNacgtrykmswbdhvn" * 4 + b"TTGACCA"], [b"NNNNACGTRYKMSWBDHVNTTGACCAXX*-"],
            [b"acgtrykmswbdhvnACGTRYKMSWBDHVN" * 3]]
    eng = mumemto_amd.Engine(0)
    for revcomp in (True, False):
        eng.set_docs(docs)
        eng.run(min_match_len=5, num_distinct=2, max_doc_freq=4, use_revcomp=revcomp)
        text, _ = O.build_text(docs, revcomp)
        assert np.array_equal(eng.text(), text)
        assert eng.output_text() == O.run(docs, min_len=5, num_distinct=2, max_doc_freq=4, revcomp=revcomp).text()
    # bytes <= 0x02 are reserved by the parse: the automatic producer falls back to the direct sort
    docs2 = [[b"ACGT\x01\x02ACGTTGCA" * 5], [b"ACGTTGCA\x01ACGT" * 4]]
    eng.set_producer("auto")
    eng.set_docs(docs2)
    eng.run(min_match_len=4, num_distinct=2, max_doc_freq=3)
    assert eng.producer_used() == "direct"       # (also under MMT_PACKED_TEXT=1: such a text keeps one byte per character)
    assert eng.output_text() == O.run(docs2, min_len=4, num_distinct=2, max_doc_freq=3).text()
    eng.set_producer("pfp")
    with pytest.raises(mumemto_amd.MumemtoError, match="reserves"):
        eng.run(min_match_len=4)
    eng.close()


def test_adversarial_shapes_against_the_oracle():
    """Homopolymers, exact copies, tandem / periodic repeats, two-letter texts, N runs: buckets of the suffix sort that
    never split early (bitonic and segmented-sort paths of the doubling rounds), matches of tens of thousands of
    characters (k_long_lcp), phrases without triggers.  Both producers against the oracle (tests/stress_shapes.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "stress_shapes.py"), "40000"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
