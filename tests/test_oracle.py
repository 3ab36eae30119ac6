"""CPU tests of the oracle itself: the checker must be right before it checks
anything.  (1) known-answer vectors recorded from the reference;
(2) brute-force definition checker on random tiny inputs, all modes;
(3) SA/LCP/BWT stream against a naive sort."""
import json
import os

import numpy as np
import pytest

import pyoracle as O
from bruteforce import bruteforce_lines
from mumemto_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_known_answer_vectors():
    spec = json.load(open(os.path.join(HERE, "golden", "toy_vectors.json")))
    for v in spec["vectors"]:
        docs = [[r.encode() for r in d] for d in v["docs"]]
        r = O.run(docs, min_len=v["min_len"], revcomp=v["revcomp"], max_doc_freq=v["max_doc_freq"])
        assert r.text() == v["expect"].encode(), v


def test_stream_against_naive_sort():
    rng = np.random.default_rng(5)
    for trial in range(30):
        n = int(rng.integers(1, 200))
        text = rng.choice(np.frombuffer(b"$ACGTN", np.uint8), size=n).astype(np.uint8)
        sa, lcp, bwt = O.build_stream(text)
        t = text.tobytes()
        order = sorted(range(n), key=lambda i: t[i:])
        assert sa[0] == n and list(sa[1:]) == order
        assert lcp[0] == 0 and lcp[1] == 0
        for j in range(2, n + 1):
            a, b = t[sa[j - 1]:], t[sa[j]:]
            k = 0
            while k < min(len(a), len(b)) and a[k] == b[k]:
                k += 1
            assert lcp[j] == k
        for j in range(n + 1):
            assert bwt[j] == (t[sa[j] - 1] if sa[j] > 0 else 0)


def _random_docs(rng, n_docs):
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(25, 60))).astype(np.uint8)
    docs = []
    for _ in range(n_docs):
        s = base.copy()
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, len(s)))] = rng.choice(np.frombuffer(b"ACGTN", np.uint8))
        if rng.random() < 0.3:  # reverse-complement a doc so '-' strand rows appear
            comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}
            s = np.array([comp[c] for c in s[::-1]], np.uint8)
        if rng.random() < 0.3:
            s = np.concatenate([s, s[: int(rng.integers(5, 20))]])
        docs.append([s.tobytes()])
    return docs


@pytest.mark.parametrize("mode", ["mum", "partial", "mem", "mem_capped", "mem_unlimited"])
def test_scan_against_bruteforce(mode):
    rng = np.random.default_rng({"mum": 1, "partial": 2, "mem": 3, "mem_capped": 4, "mem_unlimited": 6}[mode])
    for trial in range(25):
        n_docs = int(rng.integers(2, 5))
        docs = _random_docs(rng, n_docs)
        revcomp = bool(rng.integers(0, 2))
        min_len = int(rng.integers(3, 9))
        nd, f, F = {"mum": (n_docs, 1, 0), "partial": (max(2, n_docs - 1), 1, 0), "mem": (n_docs, 2, 0),
                    "mem_capped": (2, 3, n_docs + 1), "mem_unlimited": (2, 0, 0)}[mode]
        text, doc_start = O.build_text(docs, revcomp)
        sa, lcp, bwt = O.build_stream(text)
        got = O.scan(sa, lcp, bwt, doc_start, min_len=min_len, num_distinct=nd, max_doc_freq=f,
                     max_total_freq=F, revcomp=revcomp).text()
        want = bruteforce_lines(text, list(doc_start), min_len, nd, f, F, revcomp)
        assert got == want, (mode, trial, docs, revcomp, min_len)


@pytest.mark.parametrize("w,p", [(10, 100), (10, 30), (4, 7), (3, 4), (14, 50)])
def test_stream_by_way_of_the_prefix_free_parse_equals_the_suffix_sort(w, p):
    """The oracle restates both routes of the reference to the SA / LCP / BWT stream: a suffix sort of the whole text
    (`-g`, include/direct_gsacak.hpp) and the default one through the prefix-free parse (newscan.hpp -> dictionary.hpp /
    parse.hpp -> pfp.hpp -> pfp_lcp_mum.hpp:115-231).  All suffixes are distinct, so the stream cannot depend on the
    route, the window or the modulus: pangenomes with an inversion, real-assembly content (satellite
    arrays, microsatellites, runs of N, indels), exact copies, tandem repeats and one-base documents."""
    cases = [synth.pangenome(5, 4000, 0.01, seed=11, inversion=(2, 700, 1500)),
             [[b.tobytes()] for _, b in synth.haplotypes_realistic(6, 30000, 0.002, 5)],
             [[b"ACGTACGTACGTACGTNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNACGT"], [b"ACGTACGTACGTTTTTTTTTTTTTTTTTTTTTTTTT"], [b"A"],
              [b"GATTACA" * 40], [b"GATTACA" * 40], [b"acgtnryk"]]]
    for docs in cases:
        text, _ = O.build_text(docs)
        sa, lcp, bwt = O.build_stream(text)
        sa2, lcp2, bwt2, stats = O.build_stream_pfp(text, w, p)
        assert stats[3] == len(text) + 1 and stats[1] <= stats[0]
        assert np.array_equal(sa, sa2) and np.array_equal(lcp, lcp2) and np.array_equal(bwt, bwt2)
    # and the whole job (scan + writer behind either route) gives the same bytes
    docs = cases[0]
    assert O.run_job_timed(docs)[2] == O.run_job_timed(docs, pfp=(w, p))[2]


def test_cli_param_normalisation():
    # include/pfp_mum.hpp:149-198
    assert O.cli_params(16) == (16, 1, 16)
    assert O.cli_params(94, k=-1, f=3) == (93, 3, 282)
    assert O.cli_params(5, k=1) == (2, 1, 5)
    assert O.cli_params(5, k=9) == (5, 1, 5)
    assert O.cli_params(5, k=-9) == (2, 1, 5)
    assert O.cli_params(5, f=0, k=2, F=100) == (2, 0, 100)
    assert O.cli_params(5, f=2, F=100) == (5, 2, 10)
    assert O.cli_params(5, f=2, F=-1) == (5, 2, 4)
    assert O.cli_params(5, f=0, F=1) == (5, 0, 0)


def test_thresholds_and_accepted_lists():
    docs = synth.pangenome(4, 3000, 0.02, seed=11, inversion=(2, 500, 900))
    r = O.run(docs, merge=True)
    th = r.thresh()
    assert len(th) == 2 * (3000 + 1)
    acc, iv = r.accepted(), r.intervals()
    assert len(acc) >= len(iv) > 0
    # every emitted interval is also an accepted candidate, same order
    keys = {tuple(x) for x in acc.tolist()}
    assert all(tuple(x) in keys for x in iv.tolist())
    # emitted intervals come out in (closing j asc, length desc) order
    k = [(x[3], -x[2]) for x in iv.tolist()]
    assert k == sorted(k)


def test_bumbl_bytes_match_the_reference_python_writer():
    # fixtures by tests/golden/make_golden.py: the same rows written by the REFERENCE's
    # mumemto/utils.py (MUMdata.write_bums) and parsed back by it
    G = os.path.join(HERE, "golden", "bumbl")
    cases = {"strict": (dict(n_haps=5, length=3000, divergence=0.01, seed=21, inversion=(2, 500, 900)), {}),
             "partial": (dict(n_haps=5, length=3000, divergence=0.02, seed=22, inversion=(1, 200, 700)),
                         dict(num_distinct=3))}
    for name, (gen, kw) in cases.items():
        r = O.run(synth.pangenome(**gen), **kw)
        assert r.text() == open(os.path.join(G, name, "in.mums"), "rb").read()
        assert r.bumbl() == open(os.path.join(G, name, "ref.bumbl"), "rb").read()
        ref = np.load(os.path.join(G, name, "ref_from_bumbl.npz"))
        L, off, st = r.mum_rows()
        assert np.array_equal(ref["lengths"], L) and np.array_equal(ref["starts"], off)
        present = off >= 0
        assert np.array_equal(np.asarray(ref["strands"], bool)[present], st.astype(bool)[present])


def _longer_docs(rng, n_docs):
    """300 - 600 bases per document, with what the short cases cannot hold: a reverse-complement palindrome (a match of a
    document with its own other strand, running through the '$' between the strands when it sits at the document's end),
    a tandem duplication, a run of N, and substitutions between the documents."""
    comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(300, 600))).astype(np.uint8)
    half = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(12, 40))).astype(np.uint8)
    pal = np.concatenate([half, np.array([comp[c] for c in half[::-1]], np.uint8)])
    docs = []
    for d in range(n_docs):
        s = base.copy()
        for _ in range(int(rng.integers(2, 9))):
            s[int(rng.integers(0, len(s)))] = rng.choice(np.frombuffer(b"ACGTN", np.uint8))
        at = int(rng.integers(0, len(s) - len(pal)))
        if rng.random() < 0.5:
            s[at:at + len(pal)] = pal
        if d == 0:
            s = np.concatenate([s, pal])                       # palindrome at the end of the document
        if rng.random() < 0.4:
            a = int(rng.integers(0, len(s) - 40))
            s = np.concatenate([s[:a + 30], s[a:a + 30], s[a + 30:]])
        if rng.random() < 0.3:
            a = int(rng.integers(0, len(s) - 30))
            s[a:a + int(rng.integers(5, 25))] = ord("N")
        if rng.random() < 0.3:
            s = np.array([comp[c] for c in s[::-1]], np.uint8)
        docs.append([s.tobytes()])
    return docs


@pytest.mark.parametrize("mode", ["mum", "partial", "mem_capped", "mem_total_cap", "mem_unlimited"])
def test_scan_against_bruteforce_at_a_few_hundred_bases(mode):
    """The independent definition checker at 300 - 600 bases per document: MEMs with a per-document cap (-f), with a total
    cap (-F), without any cap; partial and strict multi-MUMs -- reverse-complement palindromes, duplications, runs of N."""
    rng = np.random.default_rng({"mum": 11, "partial": 12, "mem_capped": 13, "mem_total_cap": 14, "mem_unlimited": 15}[mode])
    for trial in range(16):
        n_docs = int(rng.integers(2, 5))
        docs = _longer_docs(rng, n_docs)
        revcomp = trial % 4 != 3
        min_len = int(rng.integers(8, 20))
        nd, f, F = {"mum": (n_docs, 1, 0), "partial": (max(2, n_docs - 1), 1, 0), "mem_capped": (2, 3, 0),
                    "mem_total_cap": (2, 0, n_docs + 3), "mem_unlimited": (2, 0, 0)}[mode]
        text, doc_start = O.build_text(docs, revcomp)
        sa, lcp, bwt = O.build_stream(text)
        got = O.scan(sa, lcp, bwt, doc_start, min_len=min_len, num_distinct=nd, max_doc_freq=f,
                     max_total_freq=F, revcomp=revcomp).text()
        want = bruteforce_lines(text, list(doc_start), min_len, nd, f, F, revcomp)
        assert got == want, (mode, trial, revcomp, min_len)
        assert mode != "mem_unlimited" or got.count(b"\n") > 3


def test_fast_generators_yield_the_haplotypes_numpy_would():
    """synth.ancestor_codes / uniform_codes read the generator's raw 64-bit outputs instead of asking numpy for one bounded byte at
    a time (whole-genome collections: minutes -> seconds); every fixture and digest depends on the haplotypes staying what
    `rng.integers(0, 4, size=L, dtype=uint8)` gives -- values and generator state."""
    import numpy as np
    from mumemto_amd import synth
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for seed in (1, 4, [4, 0xA11CE]):
        for L in (1, 3, 4, 5, 8, 9, 1003, 100_001):
            want = np.random.default_rng(seed).integers(0, 4, size=L, dtype=np.uint8)
            assert np.array_equal(synth.ancestor_codes(seed, L), want)
            r1, r2 = np.random.default_rng(seed), np.random.default_rng(seed)
            assert np.array_equal(synth.uniform_codes(r2, L), r1.integers(0, 4, size=L, dtype=np.uint8))
            assert r1.integers(0, 1 << 40) == r2.integers(0, 1 << 40) and r1.integers(0, 99) == r2.integers(0, 99)
            assert np.array_equal(synth.ascii_of_codes(want), acgt[want])
    # a haplotype the long way round: ancestor, Binomial(L, d) substitutions at uniform positions, the last write wins
    L, d, seed = 200_003, 0.01, 4
    anc = np.random.default_rng(seed).integers(0, 4, size=L, dtype=np.uint8)
    flat, lens = synth.collection_sparse(9, L, d, seed, which=[0, 5, 8], threads=3)
    for k, (h, seq) in enumerate(synth.haplotypes_sparse(9, L, d, seed, which=[0, 5, 8])):
        hrng = np.random.default_rng([seed, h + 1])
        want = acgt[anc]
        n = int(hrng.binomial(L, d))
        pos = hrng.integers(0, L, size=n)
        want[pos] = acgt[(anc[pos] + hrng.integers(1, 4, size=n, dtype=np.uint8)) & 3]
        assert np.array_equal(seq, want) and np.array_equal(flat[k * L:(k + 1) * L], want)
    big = np.arange(3 << 20, dtype=np.uint8)
    out = np.empty_like(big)
    synth.copy_threaded(out, big, threads=3, piece=1 << 20)
    assert np.array_equal(out, big)
    cb, cl = synth.collection_realistic(5, 300_000, 0.001, 3, which=[1, 4], procs=2)
    seqs = [s for _, s in synth.haplotypes_realistic(5, 300_000, 0.001, 3, which=[1, 4])]
    assert np.array_equal(cb, np.concatenate(seqs)) and list(cl) == [len(s) for s in seqs]
