"""GPU box helper: stage times of the partial multi-MEM mode of BASELINE C5 (-k -1 -f 3) on a synthetic collection."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth
haps = int(sys.argv[1]) if len(sys.argv) > 1 else 94
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
docs = synth.pangenome_subset(haps, L, 0.001, 3, list(range(haps)))
eng = mumemto_amd.Engine(0)
eng.set_docs(docs)
names = ["text", "suffix_sort", "lcp_bwt", "scan_kernel", "verify", "rows", "format", "total"]
for label, kw in [("strict MUM", dict(num_distinct=0, max_doc_freq=1, max_total_freq=0)),
                  ("C5: -k -1 -f 3", dict(num_distinct=haps - 1, max_doc_freq=3, max_total_freq=3 * haps)),
                  ("-k -1", dict(num_distinct=haps - 1, max_doc_freq=1, max_total_freq=0)),
                  ("MEM -f 0 -F 200", dict(num_distinct=2, max_doc_freq=0, max_total_freq=200))]:
    for rep in range(2):
        t = time.perf_counter(); eng.run(min_match_len=20, **kw); dt = time.perf_counter() - t
    ms = eng.stage_ms()
    print("%-18s %.1f ms  %s  rows %d, candidates %d, output %d bytes" % (label, dt * 1e3,
          {k: round(v, 2) for k, v in zip(names, ms)}, eng.L.mmt_num_rows(eng.h), eng.L.mmt_num_candidates(eng.h), eng.output_size()), flush=True)
