"""Brute-force definition checker (independent of the stack scan): enumerates
substrings of the text for inputs of a few hundred characters and applies the
multi-MUM / multi-MEM definition directly (SURVEY.md 8(a) note after row A9):

 * candidate = string alpha, |alpha| >= min_len, whose occurrence set (all
   text positions, '$' is an ordinary character) has >= 2 elements and is
   right-maximal (the characters following the occurrences are not all equal;
   the end of the text counts as a unique character);
 * #occ >= num_distinct, (#occ <= max_total_freq unless 0), every document
   holds <= max_doc_freq occurrences (0 = unlimited), #documents >= num_distinct;
 * left-maximal: the characters preceding the occurrences are not all equal
   (text position 0 is preceded by the 0 byte);
 * a candidate whose suffix-array interval is the last one (contains the
   lexicographically largest suffix) is never closed and is dropped
   (no flush: include/pfp_lcp_mum.hpp:223-230);
 * rows come out longest-extension-first, i.e. sorted by alpha with a proper
   extension ordered before its prefix;
 * writer rules of include/mem_finder.hpp:357-428 / 210-263.
"""


def _occurrences(text, min_len):
    """Every substring of at least min_len characters that occurs at least twice, with its occurrences.  Substrings
    are grown one character at a time and only while they still have two occurrences (a string that occurs once has no
    extension that occurs twice), which keeps a few thousand characters tractable."""
    n = len(text)
    table = {}
    level = {}
    for i in range(n - min_len + 1):
        level.setdefault(text[i:i + min_len], []).append(i)
    ln = min_len
    while level:
        nxt = {}
        for alpha, occ in level.items():
            if len(occ) < 2:
                continue
            table[alpha] = occ
            for i in occ:
                if i + ln < n:
                    nxt.setdefault(text[i:i + ln + 1], []).append(i)
        level = nxt
        ln += 1
    return table


def bruteforce_lines(text, doc_start, min_len, num_distinct, max_doc_freq, max_total_freq, revcomp):
    text = bytes(text)
    n = len(text)
    n_docs = len(doc_start) - 1
    half = [(doc_start[d + 1] - doc_start[d]) // (2 if revcomp else 1) for d in range(n_docs)]
    mummode = max_doc_freq == 1
    largest_suffix = max(range(n), key=lambda i: text[i:]) if n else -1

    def doc_of(p):
        d = 0
        while d + 1 < n_docs and doc_start[d + 1] <= p:
            d += 1
        return d

    rows = []
    for alpha, occ in _occurrences(text, min_len).items():
        if len(occ) < 2:
            continue
        ln = len(alpha)
        nxt = set()
        for p in occ:
            nxt.add(text[p + ln] if p + ln < n else ("end", p))
        if len(nxt) < 2:
            continue
        cnt = len(occ)
        if cnt < num_distinct or (max_total_freq and cnt > max_total_freq):
            continue
        per = {}
        for p in occ:
            per[doc_of(p)] = per.get(doc_of(p), 0) + 1
        if max_doc_freq and max(per.values()) > max_doc_freq:
            continue
        if len(per) < num_distinct:
            continue
        if text[largest_suffix:].startswith(alpha):
            continue  # interval is the last one of the suffix array: never closed
        prv = set(text[p - 1] if p > 0 else 0 for p in occ)
        if len(prv) < 2:
            continue
        rows.append((alpha, sorted(occ, key=lambda p: text[p:])))
    rows.sort(key=lambda r: tuple(r[0]) + (256,))

    lines = []
    for alpha, occ in rows:
        ln = len(alpha)
        if mummode:
            off = [None] * n_docs
            st = [None] * n_docs
            drop = False
            for p in occ:
                d = doc_of(p)
                cur = p - doc_start[d]
                if revcomp and cur >= half[d]:
                    if cur + ln >= 2 * half[d]:
                        drop = True
                        break
                    off[d], st[d] = 2 * half[d] - cur - ln - 1, "-"
                else:
                    off[d], st[d] = cur, "+"
            if drop:
                continue
            i = 0
            while i < n_docs - 1 and st[i] is None:
                i += 1
            if st[i] == "-":
                continue
            o = ",".join("" if off[d] is None else str(off[d]) for d in range(n_docs))
            s = ",".join("" if st[d] is None else st[d] for d in range(n_docs))
            lines.append("%d\t%s\t%s\n" % (ln, o, s))
        else:
            pos, ds, ss = [], [], []
            for k, p in enumerate(occ):
                d = doc_of(p)
                cur = p - doc_start[d]
                if revcomp and cur >= half[d]:
                    cur = 2 * half[d] - cur - ln - (0 if k == len(occ) - 1 else 1)
                    ss.append("-")
                else:
                    ss.append("+")
                pos.append(str(int(cur) % (1 << 64)))  # size_t arithmetic in write_mem (mem_finder.hpp:229,248)
                ds.append(str(d))
            lines.append("%d\t%s\t%s\t%s\n" % (ln, ",".join(pos), ",".join(ds), ",".join(ss)))
    return "".join(lines).encode()
