"""GPU box helper: two haplotypes with merge metadata (what a whole-genome partition runs): stage times, candidates.
usage: mm_probe.py <length> [haps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth

length = int(sys.argv[1]); haps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bases = np.empty(haps * length, np.uint8)
for h, b in synth.haplotypes_sparse(haps, length, 0.001, 11):
    bases[h * length:(h + 1) * length] = b
eng = mumemto_amd.Engine(0)
docs = [[bases[h * length:(h + 1) * length].tobytes()] for h in range(haps)]
eng.set_docs(docs)
for mm in (False, True, True):
    t = time.perf_counter()
    eng.run(merge_metadata=mm)
    print("merge_metadata %s: %.3f s, stage ms %s, candidates %d, rows %d, scan ranges %d" % (
        mm, time.perf_counter() - t, [round(x, 1) for x in eng.stage_ms()], eng.L.mmt_num_candidates(eng.h) if hasattr(eng.L, "mmt_num_candidates") else -1,
        eng.L.mmt_num_rows(eng.h), eng.scan_ranges()), flush=True)
