"""GPU box helper: a collection LARGER than one 32-bit suffix array, processed automatically as
anchor partitions on one GPU.  usage: big_run.py <haps> <length> [divergence]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth

haps, length = int(sys.argv[1]), int(sys.argv[2])
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
t = time.perf_counter()
docs = synth.pangenome_subset(haps, length, div, 7, list(range(haps)))
print("generated %d x %d bp in %.1f s; text = %.2f G chars" % (haps, length, time.perf_counter() - t, 2 * haps * (length + 1) / 1e9), flush=True)
eng = mumemto_amd.Engine(0)
for rep in range(2):      # the first pass pays every hipMalloc
    t = time.perf_counter()
    parts = eng.run_partitioned(docs)
    dt = time.perf_counter() - t
    print("pass %d: %.2f s, stage ms %s, pfp %s %s" % (rep, dt, [round(x, 1) for x in eng.stage_ms()],
          eng.pfp_counts(), [round(x, 1) for x in eng.pfp_stage_ms()]), flush=True)
L, off, st = eng.rows_mum()
engine_s = eng.stage_ms()[7] / 1e3      # inside libmumemto: uploads, partitions, fold, formatting, downloads
print("partitions %d, %.2f s wall in Python (%.3f Gbp/s), %.2f s inside the engine incl. H2D (%.3f Gbp/s), rows %d, "
      "output %d bytes, stage ms %s" % (parts, dt, haps * length / dt / 1e9, engine_s, haps * length / engine_s / 1e9, len(L),
                                        eng.output_size(), [round(x, 1) for x in eng.stage_ms()]), flush=True)
rng = np.random.default_rng(0)
comp = bytes.maketrans(b"ACGT", b"TGCA")
for r in rng.integers(0, len(L), size=min(300, len(L))):
    seqs = set()
    for d in range(haps):
        s = docs[d][0][off[r, d]: off[r, d] + L[r]]
        if not st[r, d]:
            s = s[::-1].translate(comp)
        seqs.add(s)
    assert len(seqs) == 1 and len(next(iter(seqs))) == L[r], r
text = eng.output_text().split(b"\n")[:-1]
anchor = docs[0][0]
keys = [anchor[int(l.split(b"\t")[1].split(b",")[0]):][: int(l.split(b"\t")[0])] for l in text[:20000]]
assert keys == sorted(keys), "rows are not in lexicographic order of the match"
print("sampled rows are real matches in every document; rows are in direct-run (lexicographic) order")
