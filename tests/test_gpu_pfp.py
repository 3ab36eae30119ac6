"""GPU parity of the PFP producer (rows A2-A4):
 A2  .dict / .parse bytes against the REAL reference parser (oracle/_ref/newscan_ref,
     fixtures tests/golden/newscan, made by tests/golden/make_golden.py);
 A3/A4  the suffix array / LCP / BWT produced through the parse against the oracle and
     against the direct producer, and end-to-end output bytes."""
import os

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth
from conftest import producer_is, PACKED_TEXT

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "newscan")


@pytest.fixture(scope="module")
def engine():
    import mumemto_amd
    e = mumemto_amd.Engine(0)
    yield e
    e.close()


def docs_of_fixture(case):
    """input.txt holds 'F rec' / 'F $' / 'R rec' lines in build_input_file_lib order."""
    docs, cur = [], []
    for line in open(os.path.join(G, case, "input.txt"), "rb").read().split(b"\n"):
        if line.startswith(b"F $"):
            if cur:
                docs.append(cur)
                cur = []
        elif line.startswith(b"F "):
            cur.append(line[2:])
    return docs


@pytest.mark.parametrize("case", sorted(os.listdir(G)))
def test_dict_and_parse_match_reference_parser(engine, case):
    w, p = map(int, open(os.path.join(G, case, "params.txt")).read().split())
    docs = docs_of_fixture(case)
    engine.set_docs(docs)
    d, q = engine.parse_only(True, w, p)
    assert d == open(os.path.join(G, case, "out.dict"), "rb").read()
    assert q.tobytes() == open(os.path.join(G, case, "out.parse"), "rb").read()


CASES = {
    "snp": dict(n_haps=5, length=20000, divergence=0.01, seed=1),
    "indel_inv": dict(n_haps=6, length=30000, divergence=0.01, seed=2, indel_rate=0.002, inversion=(2, 4000, 8000)),
    "tandem": dict(n_haps=4, length=15000, divergence=0.005, seed=3, tandem=(1, 3000, 3400, 5)),
    "n_run_lower": dict(n_haps=4, length=12000, divergence=0.02, seed=4, n_run=(0, 2000, 2600), lowercase_frac=0.1),
    "identical": dict(n_haps=3, length=5000, divergence=0.0, seed=5),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("wp", [(10, 100), (4, 11), (6, 37)])
def test_stream_through_the_parse(engine, case, wp):
    docs = synth.pangenome(**CASES[case])
    for revcomp in (True, False):
        engine.set_producer("pfp", *wp)
        engine.set_docs(docs)
        engine.run(use_revcomp=revcomp, merge_metadata=True)
        assert producer_is(engine, "pfp")
        text, doc_start = O.build_text(docs, revcomp)
        sa, lcp, bwt = O.build_stream(text)
        assert np.array_equal(engine.sa().astype(np.int64), sa[1:])
        assert np.array_equal(engine.lcp().astype(np.int64), lcp[1:])
        assert np.array_equal(engine.bwt(), bwt[1:])
        want = O.run(docs, revcomp=revcomp, merge=True)
        assert engine.output_text() == want.text()
        assert np.array_equal(engine.thresholds(), want.thresh())
    engine.set_producer("auto")


@pytest.mark.parametrize("wp", [(10, 100), (4, 11), (6, 37)])
def test_parse_statistics_equal_the_oracles_parse(engine, wp):
    """Phrases, distinct phrases and dictionary bytes of the GPU parse against the oracle's restatement of the reference's
    default route (newscan.hpp parse -> dictionary -> emitter, `mmo_build_stream_pfp`), whose stream the CPU suite holds
    against the whole-text suffix sort."""
    for case in sorted(CASES):
        docs = synth.pangenome(**CASES[case])
        engine.set_producer("pfp", *wp)
        engine.set_docs(docs)
        engine.run()
        counts = engine.pfp_counts()
        text, _ = O.build_text(docs, True)
        stats = O.build_stream_pfp(text, *wp)[3]
        # (a packed text -- MMT_PACKED_TEXT=1 -- goes through the bucket-wise producer, which builds no dictionary text)
        k = 2 if PACKED_TEXT else 3
        assert (counts["phrases"], counts["distinct"], counts["dict_len"])[:k] == stats[:k], (case, counts, stats)
    engine.set_producer("auto")


def test_degenerate_inputs_through_the_parse(engine):
    for docs in ([[b"A"], [b"A"]], [[b""], [b""]], [[b"ACGT" * 30], [b"ACGT" * 30]], [[b"A" * 3000], [b"A" * 2500]],
                 [[b"N" * 5000 + b"ACGTGGA" * 5], [b"ACGTGGA" * 5 + b"N" * 4000]]):
        for wp in [(10, 100), (3, 5)]:
            engine.set_producer("pfp", *wp)
            engine.set_docs(docs)
            engine.run(min_match_len=4)
            text, _ = O.build_text(docs, True)
            sa, lcp, bwt = O.build_stream(text)
            assert np.array_equal(engine.sa().astype(np.int64), sa[1:]), (docs[0][0][:10], wp)
            assert engine.output_text() == O.run(docs, min_len=4).text()
    engine.set_producer("auto")


def test_oversized_groups_take_the_segmented_fallback(engine):
    # long tandem repeats: some phrase suffix occurs more often than an LDS tile holds
    unit = b"ACGTTGCATTAGCCAGT"
    rnd = synth.pangenome(2, 250000, 0.3, seed=77)       # many distinct phrases sharing short trigger windows
    for docs, wps in (([[unit * 3000 + b"TTGACCA"], [b"GGA" + unit * 2500]], [(10, 100), (6, 20), (4, 11)]),
                      (rnd, [(4, 11), (3, 7)])):
        _check_fallback(engine, docs, wps)
    engine.set_producer("auto")


def _check_fallback(engine, docs, wps):
    saw_fallback = False
    for wp in wps:
        engine.set_producer("pfp", *wp)
        engine.set_docs(docs)
        engine.run(min_match_len=20, max_doc_freq=0, num_distinct=2, max_total_freq=50)
        saw_fallback |= engine.pfp_counts()["oversized_groups"] > 0
        text, _ = O.build_text(docs, True)
        sa, lcp, bwt = O.build_stream(text)
        assert np.array_equal(engine.sa().astype(np.int64), sa[1:])
        assert np.array_equal(engine.bwt(), bwt[1:])
        assert engine.output_text() == O.run(docs, min_len=20, max_doc_freq=0, num_distinct=2, max_total_freq=50).text()
    assert saw_fallback or PACKED_TEXT          # (the segmented fallback belongs to the emitter of the parse proper)


def test_parse_is_the_same_as_a_cpu_restatement_on_bigger_input(engine):
    # independent check of trigger positions: plain-python Karp-Rabin of newscan.hpp:106-114
    docs = synth.pangenome(3, 40000, 0.01, seed=9)
    engine.set_docs(docs)
    d, q = engine.parse_only(True, 10, 100)
    text, _ = O.build_text(docs, True)
    prime, w, p = 1999999973, 10, 100
    v = b"\x02" + text.tobytes() + b"\x02" * w
    h, cuts = 0, []
    win = [0] * w
    pot = pow(256, w - 1, prime)
    for i, c in enumerate(text.tobytes()):
        h = (h + prime - (win[i % w] * pot) % prime) % prime
        h = (h * 256 + c) % prime
        win[i % w] = c
        if h % p == 0 and i + 1 >= w:
            cuts.append(i)
    starts = [0] + [c - w + 2 for c in cuts]
    ends = [c + 1 for c in cuts] + [len(text) + w]
    phrases = [v[a:b + 1] for a, b in zip(starts, ends)]
    uniq = sorted(set(phrases))
    assert d == b"".join(x + b"\x01" for x in uniq) + b"\x00"
    rank = {x: i + 1 for i, x in enumerate(uniq)}
    assert list(q) == [rank[x] for x in phrases]
