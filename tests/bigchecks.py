"""Size-independent checks of a run too large for the CPU oracle: the suffix array is a permutation (count, sum,
sum of squares, xor against the closed forms), sampled neighbours are in suffix order with exactly the reported LCP and
BWT byte, sampled MUM rows are real, maximal, one-per-document matches."""
import numpy as np

_COMP = np.zeros(256, np.uint8)
_COMP[:] = np.arange(256)
for a, b in zip(b"ACGTUBDHKMRVY", b"TGCAAVHDMKYBR"):
    _COMP[a] = b


def host_text(bases, lens):
    """T = F $ revcomp(F) $ per document (src/ref_builder.cpp:211-314), as a uint8 array."""
    n = int(sum(2 * (int(l) + 1) for l in lens))
    t = np.empty(n + 64, np.uint8)
    t[n:] = 0
    at, src = 0, 0
    for l in lens:
        l = int(l)
        f = bases[src:src + l]
        t[at:at + l] = f
        t[at + l] = 36
        t[at + l + 1:at + 2 * l + 1] = _COMP[f[::-1]]
        t[at + 2 * l + 1] = 36
        at += 2 * l + 2
        src += l
    return t, n


def _lcp_of(text, n, p, q, cap=1 << 22):
    h = 0
    room = n - max(p, q)
    step = 4096
    while h < room:
        k = min(step, room - h)
        a, b = text[p + h:p + h + k], text[q + h:q + h + k]
        d = np.nonzero(a != b)[0]
        if len(d):
            return h + int(d[0])
        h += k
        step = min(step * 4, cap)
    return room


def check_stream(eng, bases, lens, samples=400, light=False, seed=0):
    text, n = host_text(bases, lens)
    assert eng.text_length() == n
    sa = eng.sa()
    assert len(sa) == n
    if not light:
        # permutation of 0..n-1: closed forms of sum, sum of squares (mod 2^64) and xor; plus min / max
        s1 = int(np.sum(sa, dtype=np.uint64))          # wraps mod 2^64, like the closed form below
        s2 = 0
        x = 0
        for a in range(0, n, 1 << 27):
            c = sa[a:a + (1 << 27)].astype(np.uint64)
            s2 = (s2 + int(np.sum(c * c, dtype=np.uint64))) & (2 ** 64 - 1)
            x ^= int(np.bitwise_xor.reduce(c))
        exp1 = (n * (n - 1) // 2) & (2 ** 64 - 1)
        exp2 = ((n - 1) * n * (2 * n - 1) // 6) & (2 ** 64 - 1)
        m = (n - 1) & 3
        expx = [n - 1, 1, n, 0][m]
        assert int(sa.min()) == 0 and int(sa.max()) == n - 1
        assert s1 == exp1 and s2 == exp2 and x == expx, "suffix array is not a permutation of the text positions"
    lcp = eng.lcp()
    bwt = eng.bwt()
    assert lcp[0] == 0
    rng = np.random.default_rng(seed)
    js = np.concatenate([rng.integers(1, n, size=samples), np.arange(1, min(n, 50)), np.arange(max(1, n - 50), n)])
    if n > 2 ** 32:      # neighbours of the first entries whose positions need the high byte
        hi = np.nonzero(sa[: min(n, 1 << 24)] >= 2 ** 32)[0][:50]
        js = np.concatenate([js, hi[hi > 0]])
    for j in js:
        j = int(j)
        p, q = int(sa[j]), int(sa[j - 1])
        h = _lcp_of(text, n, p, q)
        assert int(lcp[j]) == h, (j, p, q, int(lcp[j]), h)
        # suffix q < suffix p: the first differing character decides; the shorter suffix is smaller when one ends
        if max(p, q) + h < n:
            assert text[q + h] < text[p + h], (j, p, q, h)
        else:
            assert q > p, (j, p, q, h)
        assert int(bwt[j]) == (int(text[p - 1]) if p else 0), j
    print("stream: suffix array is a permutation; %d sampled entries are in suffix order with the reported LCP / BWT" % len(js),
          flush=True)
    return text, n


def check_mum_rows(eng, bases, lens, samples=300, seed=1, use_text=True):
    """Sampled rows: the same string at the reported offset / strand of every document, not extendable to the left or
    right in all documents at once, and rows in lexicographic order of the match."""
    L, off, st = eng.rows_mum()
    N = len(lens)
    assert len(L) > 0
    starts = np.concatenate([[0], np.cumsum(np.asarray(lens, np.uint64))]).astype(np.int64)
    rng = np.random.default_rng(seed)

    def occ(d, o, ln, plus):
        f = bases[starts[d]:starts[d + 1]]
        if plus:
            return f[o:o + ln], (f[o - 1] if o > 0 else 36), (f[o + ln] if o + ln < len(f) else 36)
        # '-' strand: the reported offset is the forward coordinate of the segment whose reverse complement matches
        # (write_mum, mem_finder.hpp:365-380: pos = 2 (L + 1) - pos - len - 1)
        seg = _COMP[f[o:o + ln][::-1]]
        left = _COMP[f[o + ln]] if o + ln < len(f) else 36
        right = _COMP[f[o - 1]] if o > 0 else 36
        return seg, left, right
    for r in rng.integers(0, len(L), size=min(samples, len(L))):
        ln = int(L[r])
        segs, lefts, rights = [], set(), set()
        for d in range(N):
            assert off[r, d] >= 0
            s, a, b = occ(d, int(off[r, d]), ln, bool(st[r, d]))
            segs.append(s.tobytes()); lefts.add(int(a)); rights.add(int(b))
        assert len(set(segs)) == 1 and len(segs[0]) == ln, r
        assert len(lefts) > 1 and len(rights) > 1, ("row is not maximal", r)
    a0 = bases[starts[0]:starts[1]]
    keys = []
    if use_text:
        text = eng.output_text().split(b"\n")[:-1]
        assert len(text) == len(L)
        for line in text[:20000]:
            f = line.split(b"\t")
            o = int(f[1].split(b",")[0])
            keys.append(a0[o:o + int(f[0])].tobytes())
    else:           # tens of millions of rows: the order from the row arrays (the anchor is '+' in every kept row)
        for r in range(min(20000, len(L))):
            assert st[r, 0] == 1
            keys.append(a0[int(off[r, 0]):int(off[r, 0]) + int(L[r])].tobytes())
    assert keys == sorted(keys), "rows are not in lexicographic order of the match"
    print("rows: %d rows; %d sampled rows are real, maximal matches in every document; order is lexicographic" % (len(L), min(samples, len(L))), flush=True)


def check_mem_rows(eng, bases, lens, min_docs, max_doc_freq, samples=300, seed=2, text=None):
    """Sampled rows of a partial multi-MUM / multi-MEM run: every listed occurrence spells the same string of the
    text T (matches may run through a '$'), the occurrences cover at least min_docs documents with at most
    max_doc_freq each, and the row is maximal."""
    L, occ, off, ids, st = eng.rows_mem()
    if text is None:
        text, n = host_text(bases, lens)
    lens = [int(l) for l in lens]
    doc_start = np.concatenate([[0], np.cumsum([2 * (l + 1) for l in lens])]).astype(np.int64)
    rng = np.random.default_rng(seed)
    assert len(L) > 0
    checked = 0
    for r in rng.integers(0, len(L), size=min(samples, len(L))):
        ln = int(L[r])
        a, b = int(occ[r]), int(occ[r + 1])
        counts = np.bincount(ids[a:b].astype(np.int64), minlength=len(lens))
        assert (counts > 0).sum() >= min_docs and counts.max() <= max_doc_freq, r
        strings, lefts, rights = set(), set(), set()
        ok = True
        for k in range(a, b):
            d, o = int(ids[k]), int(off[k])
            if st[k]:
                tp = int(doc_start[d]) + o
            else:
                # write_mem: pos = 2 (L + 1) - pos - len - 1, without the "- 1" for the last listed occurrence of the row
                # (mem_finder.hpp:229 vs :248); positions are size_t there, an occurrence that runs into the terminator
                # wraps around -- such rows are skipped here
                if k == b - 1:
                    o -= 1
                tp = int(doc_start[d]) + 2 * (lens[d] + 1) - o - ln - 1
                if o < 0 or tp < int(doc_start[d]) + lens[d] + 1:
                    ok = False
                    break
            strings.add(text[tp:tp + ln].tobytes())
            lefts.add(int(text[tp - 1]) if tp > 0 else 0); rights.add(int(text[tp + ln]))
        if not ok:
            continue
        checked += 1
        assert len(strings) == 1 and len(next(iter(strings))) == ln, (r, len(strings))
        assert len(lefts) > 1 and len(rights) > 1, ("row is not maximal", r)
    assert checked > 0
    print("rows: %d rows with %d occurrences; %d sampled rows are real, maximal matches" % (len(L), len(off), checked), flush=True)


class SparseModel:
    """mumemto_amd.synth.haplotypes_sparse as a random-access model: the ancestor + the substitutions of every haplotype, so
    that a collection that does not fit the host as bytes (94 x 3.05 Gbp = 287 GB; the boxes of this pool give a container
    300 GiB) can be SUPPLIED document by document (`fill`) and read back piecewise by the checks (`doc(h)[a:b]`)."""
    _ACGT = np.frombuffer(b"ACGT", np.uint8)

    def __init__(self, n_total, length, divergence, seed, which):
        from mumemto_amd import synth
        self.length, self.div, self.seed, self.which = length, divergence, seed, list(which)
        self.anc = synth.ancestor_codes(seed, length)
        self.anc_ascii = synth.ascii_of_codes(self.anc)
        self._subs = {}

    def _draw(self, h):
        hrng = np.random.default_rng([self.seed, h + 1])
        k = int(hrng.binomial(self.length, self.div)) if self.div > 0 else 0
        if not k:
            return np.zeros(0, np.int64), np.zeros(0, np.uint8)
        pos = hrng.integers(0, self.length, size=k)
        return pos, self._ACGT[(self.anc[pos] + hrng.integers(1, 4, size=k, dtype=np.uint8)) & 3]

    def fill(self, d, dst):
        """the bases of document d (= haplotype which[d]) into dst, exactly as haplotypes_sparse yields them"""
        from mumemto_amd import synth
        pos, val = self._draw(self.which[d])
        synth.copy_threaded(dst, self.anc_ascii)              # (3 GB a document: the supplier's time is the copy)
        dst[pos] = val

    def subs(self, d):
        if d not in self._subs:
            pos, val = self._draw(self.which[d])
            # dst[pos] = val: the last write to a position wins
            order = np.argsort(pos, kind="stable")
            pos, val = pos[order], val[order]
            last = np.ones(len(pos), bool)
            last[:-1] = pos[1:] != pos[:-1]
            self._subs[d] = (pos[last], val[last])
        return self._subs[d]

    def doc(self, d):
        return _ModelDoc(self, d)


class _ModelDoc:
    def __init__(self, model, d):
        self.m, self.d = model, d

    def __len__(self):
        return self.m.length

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b, step = key.indices(self.m.length)
            assert step == 1
            seg = self.m.anc_ascii[a:b].copy()
            pos, val = self.m.subs(self.d)
            i, j = np.searchsorted(pos, a), np.searchsorted(pos, b)
            seg[pos[i:j] - a] = val[i:j]
            return seg
        return self[int(key):int(key) + 1][0]


class LazyText:
    """T = F $ revcomp(F) $ per document, read from the bases on demand (a 250 G-character text does not get a host copy):
    text[i] and text[a:b] as numpy uint8, positions beyond the text read as 0."""

    def __init__(self, bases, lens):
        self.bases, self.lens = bases, [int(l) for l in lens]
        self.doc_start = np.concatenate([[0], np.cumsum([2 * (l + 1) for l in self.lens])]).astype(np.int64)
        self.base_start = np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int64)
        self.n = int(self.doc_start[-1])

    def _doc(self, d):
        """the bases of document d: a view of the flat array, or of the model a supplied collection was generated from"""
        if hasattr(self.bases, "doc"):
            return self.bases.doc(d)
        return self.bases[self.base_start[d]:self.base_start[d + 1]]

    def _piece(self, d, lo, hi):
        """local positions [lo, hi) of document d"""
        L = self.lens[d]
        f = self._doc(d)
        out = np.empty(hi - lo, np.uint8)
        for k, p in enumerate(range(lo, hi)):
            if p < L:
                out[k] = f[p]
            elif p == L or p == 2 * L + 1:
                out[k] = 36
            else:
                out[k] = _COMP[f[2 * L - p]]
        return out

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = key.start or 0, self.n + 64 if key.stop is None else key.stop
            out = np.zeros(max(b - a, 0), np.uint8)
            p = a
            while p < min(b, self.n):
                d = int(np.searchsorted(self.doc_start, p, side="right")) - 1
                end = min(b, int(self.doc_start[d + 1]))
                lo = p - int(self.doc_start[d])
                # long stretches inside one strand: vectorised
                L = self.lens[d]
                f = self._doc(d)
                hi = end - int(self.doc_start[d])
                if hi <= L:
                    out[p - a:end - a] = f[lo:hi]
                elif lo > L and hi <= 2 * L + 1:
                    out[p - a:end - a] = _COMP[f[2 * L - hi + 1:2 * L - lo + 1][::-1]]
                else:
                    out[p - a:end - a] = self._piece(d, lo, hi)
                p = end
            return out
        i = int(key)
        if i < 0 or i >= self.n:
            return np.uint8(0)
        return self[i:i + 1][0]


def check_mems_file(path, text, lens, min_docs, max_doc_freq, samples=200, seed=3):
    """Sampled rows of a PREFIX.mems file (LEN \\t offsets \\t documents \\t strands; include/mem_finder.hpp:210-263): the checks of
    check_mem_rows on lines read at random places of the file (a run that wrote its rows window by window keeps no arrays)."""
    import os
    size = os.path.getsize(path)
    lens = [int(l) for l in lens]
    doc_start = np.concatenate([[0], np.cumsum([2 * (l + 1) for l in lens])]).astype(np.int64)
    rng = np.random.default_rng(seed)
    checked = rows_seen = 0
    with open(path, "rb") as f:
        for at in np.sort(rng.integers(0, max(size - 1, 1), size=samples)):
            f.seek(int(at))
            f.readline()
            line = f.readline()
            if not line.endswith(b"\n"):
                continue
            rows_seen += 1
            ln, offs, ids, sts = line[:-1].split(b"\t")
            ln = int(ln); offs = [int(x) for x in offs.split(b",")]; ids = [int(x) for x in ids.split(b",")]; sts = sts.split(b",")
            assert len(offs) == len(ids) == len(sts)
            counts = np.bincount(ids, minlength=len(lens))
            assert (counts > 0).sum() >= min_docs and counts.max() <= max_doc_freq, line[:80]
            strings, lefts, rights = set(), set(), set()
            ok = True
            for k in range(len(offs)):
                d, o = ids[k], offs[k]
                if sts[k] == b"+":
                    tp = int(doc_start[d]) + o
                else:
                    if k == len(offs) - 1:
                        o -= 1
                    tp = int(doc_start[d]) + 2 * (lens[d] + 1) - o - ln - 1
                    if o < 0 or o >= 2**63 or tp < int(doc_start[d]) + lens[d] + 1:
                        ok = False
                        break
                strings.add(text[tp:tp + ln].tobytes())
                lefts.add(int(text[tp - 1]) if tp > 0 else 0); rights.add(int(text[tp + ln]))
            if not ok:
                continue
            checked += 1
            assert len(strings) == 1 and len(next(iter(strings))) == ln, line[:80]
            assert len(lefts) > 1 and len(rights) > 1, ("row is not maximal", line[:80])
    assert checked > 0
    print("file: %d bytes; %d sampled rows of %d are real, maximal matches in at least %d documents" % (size, checked, rows_seen, min_docs), flush=True)


def _text_piece(text, a, b, n):
    """text[a:b] as numpy uint8, zeros beyond the text (numpy arrays and LazyText alike)"""
    out = np.zeros(b - a, np.uint8)
    hi = min(b, n)
    if hi > a:
        out[:hi - a] = np.asarray(text[a:hi], np.uint8)
    return out


def check_bins_complete(eng, text, n_text, doc_start, kmers, min_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0,
                        revcomp=True):
    """PRECISION AND RECALL inside whole bins of a run of any size.  `eng` ran with eng.set_row_tap(kmers): every interval it
    accepted whose match begins with one of the k-mers left a copy (length + all its text positions).  Here, per k-mer: the GPU
    lists every text position whose suffix begins with it (eng.kmer_positions over the resident text), the host spells those
    suffixes from `text` (a numpy array or a LazyText over the generator's model), sorts them, computes LCP / BWT / document of
    each -- a piece of the stream: every interval with an LCP value >= len(kmer) that touches the bin lies inside it -- and runs
    the ORACLE's scan (mmo_scan: the restatement of mem_finder.hpp:161-170,304-355) over that piece with a closing entry behind
    it; the intervals it reports must be exactly the tapped ones.  Returns (bins, suffixes, rows) checked."""
    import pyoracle as O
    kmers = [bytes(k) for k in kmers]
    k = len(kmers[0])
    assert k <= min_len, "a bin's prefix must not be longer than the shortest reportable match"
    t_len, t_start, t_sa = eng.row_tap()
    got = {}
    for r in range(len(t_len)):
        occ = np.sort(t_sa[int(t_start[r]):int(t_start[r + 1])].astype(np.int64))
        first = _text_piece(text, int(occ[0]), int(occ[0]) + k, n_text).tobytes()
        got.setdefault(first, set()).add((int(t_len[r]), tuple(int(x) for x in occ)))
    pos, which = eng.kmer_positions(kmers, cap=1 << 24)
    doc_start = np.ascontiguousarray(doc_start, np.int64)
    suffixes = rows = 0
    for i, km in enumerate(kmers):
        P = pos[which == i].astype(np.int64)
        want = set()
        if len(P):
            step = 512
            while True:
                strs = [_text_piece(text, int(p), int(p) + step, n_text) for p in P]
                order = sorted(range(len(P)), key=lambda j: strs[j].tobytes())
                lcp = np.zeros(len(P) + 1, np.int64)
                again = False
                for a in range(1, len(P)):
                    x, y = strs[order[a - 1]], strs[order[a]]
                    neq = np.nonzero(x != y)[0]
                    l = int(neq[0]) if len(neq) else step
                    # (the end of the text is a unique, smallest sentinel: two suffixes never agree beyond it)
                    l = min(l, n_text - int(P[order[a - 1]]), n_text - int(P[order[a]]))
                    if l >= step:
                        again = True
                        break
                    lcp[a] = l
                if not again:
                    break
                step *= 4
            sa = np.zeros(len(P) + 1, np.int64)
            sa[:len(P)] = P[order]
            bwt = np.zeros(len(P) + 1, np.uint8)
            for a in range(len(P)):
                bwt[a] = _text_piece(text, int(sa[a]) - 1, int(sa[a]), n_text)[0] if sa[a] > 0 else 0
            # the closing entry: some suffix of another bin (LCP 0 with everything here)
            res = O.scan(sa, lcp, bwt, doc_start, min_len=min_len, num_distinct=num_distinct, max_doc_freq=max_doc_freq,
                         max_total_freq=max_total_freq, revcomp=revcomp)
            for s, e, l, _j in res.intervals():
                want.add((int(l), tuple(sorted(int(x) for x in sa[int(s):int(e) + 1]))))
        have = got.pop(km, set())
        missing, extra = want - have, have - want
        assert not missing and not extra, ("bin %r: %d positions, %d rows expected, %d tapped; missing %s; not expected %s"
                                           % (km, len(P), len(want), len(have), sorted(missing)[:2], sorted(extra)[:2]))
        suffixes += len(P); rows += len(want)
    assert not got, "tapped rows that begin with none of the k-mers: %r" % list(got)[:3]
    return len(kmers), suffixes, rows
