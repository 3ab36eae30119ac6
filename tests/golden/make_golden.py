#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run HERE (needs /root/reference
mounted and `make -C oracle` done); the GPU box only reads the fixtures.

anchor_merge/<case>/: partition inputs (p*.mums + p*.athresh, produced by the
  oracle's scan) and the outputs of the REAL reference binary
  oracle/_ref/anchor_merge (built from src/merge_candidates.cpp) on them:
  merged.mums + merged.athresh.  direct.mums is the oracle's direct run on the
  union of the partitions (used for the re-sort property).
newscan/<case>/: input record lines and the dict/parse bytes written by the
  REAL reference parser oracle/_ref/newscan_ref (include/newscan.hpp).
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyoracle as O  # noqa: E402
from mumemto_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(ROOT, "oracle", "_ref")


def anchor_case(name, docs, groups):
    d = os.path.join(OUT, "anchor_merge", name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    tmp = tempfile.mkdtemp()
    paths = []
    for gi, g in enumerate(groups):
        sub = [docs[i] for i in g]
        r = O.run(sub, merge=True)
        L0 = len(b"".join(docs[0]))
        open(os.path.join(tmp, "p%d.mums" % gi), "wb").write(r.text())
        r.thresh()[: L0 + 1].tofile(os.path.join(tmp, "p%d.athresh" % gi))
        paths.append(os.path.join(tmp, "p%d.mums" % gi))
    subprocess.check_call([os.path.join(REF, "anchor_merge")] + paths + ["-o", os.path.join(tmp, "merged")],
                          stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        shutil.copy(os.path.join(tmp, f), os.path.join(d, f))
    order = [groups[0][0]] + [i for g in groups for i in g[1:]]
    direct = O.run([docs[i] for i in order], merge=True)
    open(os.path.join(d, "direct.mums"), "wb").write(direct.text())
    direct.thresh()[: len(b"".join(docs[0])) + 1].tofile(os.path.join(d, "direct.athresh"))
    np.save(os.path.join(d, "anchor.npy"), np.frombuffer(b"".join(docs[0]), np.uint8))
    with open(os.path.join(d, "groups.txt"), "w") as f:
        f.write("\n".join(",".join(map(str, g)) for g in groups) + "\n")
    shutil.rmtree(tmp)


def newscan_case(name, lines, w=10, p=100):
    d = os.path.join(OUT, "newscan", name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    open(os.path.join(d, "input.txt"), "wb").write(b"".join(l + b"\n" for l in lines))
    subprocess.run([os.path.join(REF, "newscan_ref"), str(w), str(p), os.path.join(d, "out")],
                   input=b"".join(l + b"\n" for l in lines), check=True, stderr=subprocess.DEVNULL)
    open(os.path.join(d, "params.txt"), "w").write("%d %d\n" % (w, p))


def bumbl_case(name, docs, **kw):
    """<case>/in.mums + in.bumbl: written by the oracle; ref.bumbl / ref.mums: the same rows written
    by the REFERENCE's Python (mumemto/utils.py MUMdata.write_bums / write_mums after parsing in.mums);
    ref_from_bumbl.npz: arrays the reference parses out of in.bumbl."""
    sys.path.insert(0, "/root/reference/mumemto")
    import importlib
    utils = importlib.import_module("utils")
    d = os.path.join(OUT, "bumbl", name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    r = O.run(docs, **kw)
    open(os.path.join(d, "in.mums"), "wb").write(r.text())
    open(os.path.join(d, "in.bumbl"), "wb").write(r.bumbl())
    m = utils.MUMdata(os.path.join(d, "in.mums"), sort=False)
    m.write_bums(os.path.join(d, "ref.bumbl"))
    m.write_mums(os.path.join(d, "ref.mums"))
    b = utils.MUMdata(os.path.join(d, "in.bumbl"), sort=False)
    np.savez(os.path.join(d, "ref_from_bumbl.npz"), lengths=b.lengths, starts=b.starts, strands=b.strands)


def main():
    docs = synth.pangenome(6, 4000, 0.01, seed=7, indel_rate=0.002, inversion=(3, 1000, 1400))
    anchor_case("two_way", docs, [[0, 1, 2], [0, 3, 4, 5]])
    anchor_case("three_way", docs, [[0, 1], [0, 2, 3], [0, 4, 5]])
    docs2 = synth.pangenome(5, 2500, 0.03, seed=9, tandem=(1, 700, 760, 3))
    anchor_case("divergent", docs2, [[0, 1, 2], [0, 3, 4]])

    # PFP parser fixtures: forward + '$' + revcomp + '$' per doc, like
    # build_input_file_lib (src/ref_builder.cpp:330-384)
    d3 = synth.pangenome(3, 6000, 0.01, seed=3, lowercase_frac=0.05)
    lines = []
    for doc in d3:
        for rec in doc:
            lines.append(b"F " + rec)
        lines.append(b"F $")
        for rec in reversed(doc):
            lines.append(b"R " + rec)
        lines.append(b"F $")
    newscan_case("three_docs_w10_p100", lines)
    newscan_case("three_docs_w4_p11", lines, w=4, p=11)
    bumbl_case("strict", synth.pangenome(5, 3000, 0.01, seed=21, inversion=(2, 500, 900)))
    bumbl_case("partial", synth.pangenome(5, 3000, 0.02, seed=22, inversion=(1, 200, 700)), num_distinct=3)
    newscan_case("tiny", [b"F ACGTACGTTTGACCA", b"F $", b"R ACGTACGTTTGACCA", b"F $"], w=3, p=5)


if __name__ == "__main__":
    main()
