#!/usr/bin/env python3
"""Golden fixtures of the string-based merge.  Run HERE: needs /root/reference mounted and
`make -C oracle` done; the fixtures are data only.

string_merge/<case>/
  doc<i>.fa groups.txt            the documents and which of them form partition g (one line per partition)
  p<g>.mums .thresh .thresh_rev .lengths     partition outputs of a `-M` run (CPU oracle; rows sorted by their offset
                                  in the partition's first document, which is the order PREFIX.thresh is written in,
                                  include/mem_finder.hpp:126-130)
  p<g>_mums.fa                    written by the REAL reference tool oracle/_ref/extract_mums (src/extract_mums.cpp)
  mom.mums mom.lengths            multi-MUMs of the p<g>_mums.fa collections (CPU oracle)
  merged.mums .thresh .thresh_rev .lengths merged_bin.bumbl   written by the REAL reference merge (mumemto/merge_mums.py main(), imported
                                  from /root/reference, run with -m mom.mums)
  direct.mums                     the oracle's direct run on the union of the partitions, rows sorted by first offset
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), "/root/reference/mumemto"]
import pyoracle as O  # noqa: E402
from mumemto_amd import synth  # noqa: E402
import merge_mums as ref_merge  # noqa: E402  (the reference's module; never copied)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "string_merge")
REF = os.path.join(ROOT, "oracle", "_ref")


def write_fasta(path, records, names=None):
    with open(path, "wb") as f:
        for i, r in enumerate(records):
            f.write(b">" + (names[i] if names else b"rec%d" % i) + b"\n")
            for k in range(0, len(r), 80):
                f.write(r[k:k + 80] + b"\n")


def sort_rows_by_first_offset(text):
    rows = [ln for ln in text.split(b"\n") if ln]
    rows.sort(key=lambda ln: int(ln.split(b"\t")[1].split(b",")[0]))
    return b"\n".join(rows) + (b"\n" if rows else b"")


def lengths_lines(paths, docs):
    out = []                          # the form RefBuilder::write_lengths_file writes (src/ref_builder.cpp:193-209),
    for p, d in zip(paths, docs):     # with relative instead of canonical paths so that the fixture can move
        out.append("%s * %d" % (p, sum(len(r) for r in d)))
        out += ["%s rec%d %d" % (p, i, len(r)) for i, r in enumerate(d)]
    return "\n".join(out) + "\n"


def read_fasta_records(path):
    recs, cur = [], None
    for ln in open(path, "rb").read().split(b"\n"):
        if ln.startswith(b">"):
            if cur is not None:
                recs.append(b"".join(cur))
            cur = []
        elif ln:
            cur.append(ln)
    if cur is not None:
        recs.append(b"".join(cur))
    return recs


def case(name, docs, groups):
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    cwd = os.getcwd()
    os.chdir(d)                       # relative paths inside the .lengths files keep the fixture relocatable
    try:
        for gi, g in enumerate(groups):
            sub = [docs[i] for i in g]
            fa = []
            for i in g:
                fa.append("doc%d.fa" % i)
                write_fasta(fa[-1], docs[i])
            r = O.run(sub, merge=True)
            open("p%d.mums" % gi, "wb").write(sort_rows_by_first_offset(r.text()))
            r.thresh_file(False).tofile("p%d.thresh" % gi)
            r.thresh_file(True).tofile("p%d.thresh_rev" % gi)
            open("p%d.lengths" % gi, "w").write(lengths_lines(fa, sub))
            subprocess.check_call([os.path.join(REF, "extract_mums"), "-m", "p%d.mums" % gi])
        mom_docs = [read_fasta_records("p%d_mums.fa" % gi) for gi in range(len(groups))]
        mom = O.run(mom_docs)
        open("mom.mums", "wb").write(mom.text())
        with open("mom.lengths", "w") as f:
            for gi, recs in enumerate(mom_docs):
                f.write("p%d_mums.fa * %d\n" % (gi, sum(len(x) for x in recs)))
                for k, x in enumerate(recs):
                    f.write("p%d_mums.fa mum_%d %d\n" % (gi, k, len(x)))
        args = ref_merge.parse_arguments(["-m", "mom.mums", "-o", "merged"] + ["p%d.mums" % gi for gi in range(len(groups))])
        ref_merge.main(args)
        args = ref_merge.parse_arguments(["-m", "mom.mums", "-o", "merged_bin.bumbl"] + ["p%d.mums" % gi for gi in range(len(groups))])
        ref_merge.main(args)                      # the same rows through the reference's .bumbl writer
        for f in ("merged_bin.thresh", "merged_bin.thresh_rev", "merged_bin.lengths"):
            os.remove(f)
        with open("groups.txt", "w") as f:
            f.write("\n".join(",".join(map(str, g)) for g in groups) + "\n")
        order = [i for g in groups for i in g]
        direct = O.run([docs[i] for i in order])
        open("direct.mums", "wb").write(sort_rows_by_first_offset(direct.text()))
        same = open("merged.mums", "rb").read() == open("direct.mums", "rb").read()
        print("%s: merged == direct: %s (%d rows)" % (name, same, open("merged.mums", "rb").read().count(b"\n")))
    finally:
        os.chdir(cwd)


def main():
    docs = synth.pangenome(6, 6000, 0.01, 21, inversion=(3, 2000, 2600))
    case("two_by_two", docs[:4], [[0, 1], [2, 3]])
    case("three_parts", docs, [[0, 1], [2, 3], [4, 5]])
    case("inverted_first", docs[:4], [[0, 1], [3, 2]])     # partition 1's MUM strings match reversed inside the inversion
    # two partitions whose MUMs coincide up to one extra SNP: one MUMs-of-MUMs match spans dozens of `#` and is cut
    # back into the MUMs (identical partitions are no use: the match then runs past the end of the collection and the
    # reference's merge stops with an IndexError)
    rows = sorted((int(ln.split(b"\t")[1].split(b",")[1]), int(ln.split(b"\t")[0]))
                  for ln in O.run(docs[:2]).text().splitlines())
    c = bytearray(docs[1][0])
    for r in (rows[60], rows[-1]):                          # the middle of two MUMs, in document 1's coordinates
        at = min(r[0] + r[1] // 2, len(c) - 5)
        c[at] = ord("A") if c[at] != ord("A") else ord("C")
    case("shared_boundaries", [docs[0], docs[1], [bytes(c)]], [[0, 1], [0, 2]])


if __name__ == "__main__":
    main()
