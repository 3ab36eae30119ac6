"""CPU: the oracle's scan / writers / thresholds (oracle/mumemto_oracle.c, rows A5 - A7) against the independent definition
checker of tests/bruteforce.py, in volume: 2,000 seeded cases PER MODE (MMT_FUZZ_CASES overrides), spread over the cores
of the box.  `mem_finder.hpp` cannot be built here (sdsl-lite / gsacak are fetched from the network by the reference's
CMake and absent from the image), so the scan's pin is the eight recorded toy vectors + this checker, which shares nothing
with the stack scan: it grows every repeated substring from a dictionary of k-mers and applies the definitions of SURVEY
8(a) A5 / A6 / A7 directly.

What a case holds (seeded; the case number is the seed, a failure prints it): 2 - 5 documents of 120 - 400 bases derived
from one ancestor by substitutions, short indels, and
  * a reverse-complement palindrome at the END of a document (a match then runs through the '$' between the strands),
  * a palindrome inside, a tandem duplication, a run of N, IUPAC codes, lower-case records, a document that is the reverse
    complement of its neighbours, a document that ends in a prefix of itself;
modes: strict multi-MUMs (+ the thresholds of A7 against their definition: the longest proper prefix of the match that occurs
more often than the match), partial multi-MUMs (-k), multi-MEMs with a per-document cap (-f 2 / 3), with a total cap (-F),
without any cap (-f 0 -F 0); min_len 4 - 20; reverse complement on in three cases of four."""
import multiprocessing as mp
import os

import numpy as np
import pytest

CASES = int(os.environ.get("MMT_FUZZ_CASES", "2000"))
MODES = ["mum", "partial", "mem_f", "mem_F", "mem_unlimited"]
_COMP = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}


def make_case(seed, mode):
    rng = np.random.default_rng([seed, MODES.index(mode)])
    n_docs = int(rng.integers(2, 6))
    anc = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(120, 400))).astype(np.uint8)
    half = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(6, 16))).astype(np.uint8)
    pal = np.concatenate([half, np.array([_COMP[c] for c in half[::-1]], np.uint8)])
    docs = []
    for d in range(n_docs):
        s = anc.copy()
        for _ in range(int(rng.integers(0, 6))):
            s[int(rng.integers(0, len(s)))] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
        if rng.random() < 0.3:                                  # short indel
            a = int(rng.integers(1, len(s) - 1))
            s = np.delete(s, slice(a, a + int(rng.integers(1, 4)))) if rng.random() < 0.5 else np.insert(
                s, a, rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(1, 4))))
        if rng.random() < 0.4:
            a = int(rng.integers(0, len(s) - len(pal)))
            s[a:a + len(pal)] = pal
        if rng.random() < 0.35:                                 # tandem duplication
            a = int(rng.integers(0, max(1, len(s) - 40)))
            k = int(rng.integers(8, 36))
            s = np.concatenate([s[:a + k], s[a:a + k], s[a + k:]])
        if rng.random() < 0.25:
            a = int(rng.integers(0, max(1, len(s) - 30)))
            s[a:a + int(rng.integers(3, 25))] = ord("N")
        if rng.random() < 0.1:
            s[int(rng.integers(0, len(s)))] = rng.choice(np.frombuffer(b"RYKMSWN", np.uint8))
        if rng.random() < 0.25:
            s = np.array([_COMP.get(int(c), int(c)) for c in s[::-1]], np.uint8)
        if rng.random() < 0.3:
            s = np.concatenate([s, pal])                        # palindromic document end
        if rng.random() < 0.15:
            s = np.concatenate([s, s[: int(rng.integers(5, 30))]])
        rec = s.tobytes()
        if rng.random() < 0.15:
            rec = rec.lower()
        if rng.random() < 0.2 and len(rec) > 40:                # two records: concatenated without a separator
            cut = int(rng.integers(10, len(rec) - 10))
            docs.append([rec[:cut], rec[cut:]])
        else:
            docs.append([rec])
    revcomp = bool(rng.integers(0, 4))
    min_len = int(rng.integers(4, 21))
    nd, f, F = {"mum": (n_docs, 1, 0), "partial": (max(2, n_docs - int(rng.integers(1, 3))), 1, 0),
                "mem_f": (int(rng.integers(2, n_docs + 1)), int(rng.integers(2, 4)), 0),
                "mem_F": (2, 0, int(rng.integers(n_docs, 3 * n_docs))), "mem_unlimited": (2, 0, 0)}[mode]
    return docs, revcomp, min_len, nd, f, F


def _thresholds_by_definition(text, doc_start, min_len, n_docs):
    """A7 for strict multi-MUM candidates: for every string alpha that occurs exactly once in every document (whether or
    not it is left-maximal: the threshold is recorded BEFORE the BWT test, mem_finder.hpp:326-336), right-maximal and closed
    by a later suffix, thresh[position of its occurrence in document 0] = the length of the longest proper prefix of alpha
    with more occurrences than alpha -- max(LCP[start], LCP[end + 1]) in the reference's terms -- capped at 65535."""
    from bruteforce import _occurrences
    text = bytes(text)
    n = len(text)
    out = np.zeros(doc_start[1] - doc_start[0], np.uint16)
    largest = max(range(n), key=lambda i: text[i:]) if n else -1

    def doc_of(p):
        d = 0
        while d + 1 < n_docs and doc_start[d + 1] <= p:
            d += 1
        return d

    def count(s):
        c, at = 0, text.find(s)
        while at >= 0:
            c += 1
            at = text.find(s, at + 1)
        return c
    for alpha, occ in _occurrences(text, min_len).items():
        if len(occ) != n_docs or sorted(doc_of(p) for p in occ) != list(range(n_docs)):
            continue
        ln = len(alpha)
        if len(set(text[p + ln] if p + ln < n else ("end", p) for p in occ)) < 2:
            continue                                            # not right-maximal: not an LCP interval
        if text[largest:].startswith(alpha):
            continue                                            # the last interval is never closed
        t = 0
        for k in range(ln - 1, 0, -1):
            if count(alpha[:k]) > n_docs:
                t = k
                break
        p0 = [p for p in occ if doc_of(p) == 0][0]
        out[p0 - doc_start[0]] = min(t, 65535)
    return out


def run_chunk(args):
    mode, seeds = args
    import pyoracle as O
    from bruteforce import bruteforce_lines
    bad, rows = [], 0
    for seed in seeds:
        docs, revcomp, min_len, nd, f, F = make_case(seed, mode)
        text, doc_start = O.build_text(docs, revcomp)
        sa, lcp, bwt = O.build_stream(text)
        r = O.scan(sa, lcp, bwt, doc_start, min_len=min_len, num_distinct=nd, max_doc_freq=f, max_total_freq=F,
                   revcomp=revcomp, merge=(mode == "mum"))
        got = r.text()
        want = bruteforce_lines(text, list(doc_start), min_len, nd, f, F, revcomp)
        rows += want.count(b"\n")
        if got != want:
            bad.append((seed, "rows"))
        elif mode == "mum" and seed % 4 == 0:
            # candidate_thresh holds 2 (L_0 + 1) entries whether or not the reverse strand is in the text (mem_finder.hpp:98)
            th, want_th = r.thresh(), _thresholds_by_definition(text, list(doc_start), min_len, len(docs))
            if not np.array_equal(th[: len(want_th)], want_th) or th[len(want_th):].any():
                bad.append((seed, "thresholds"))
    return bad, rows


@pytest.mark.parametrize("mode", MODES)
def test_oracle_scan_equals_the_definition_checker_in_volume(mode):
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    seeds = list(range(CASES))
    chunks = [(mode, seeds[i::workers * 4]) for i in range(workers * 4)]
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(run_chunk, chunks)
    bad = [b for r in res for b in r[0]]
    rows = sum(r[1] for r in res)
    assert not bad, "oracle != definition checker for %s cases (seed, what): %s" % (mode, bad[:10])
    assert rows > CASES // 2, "the cases of mode %s hardly produce any rows (%d)" % (mode, rows)
