"""GPU box helper: stage times of partial multi-MUM modes (-k) on a collection of many documents."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mumemto_amd
from mumemto_amd import synth
haps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
docs = synth.pangenome_subset(haps, L, float(sys.argv[3]) if len(sys.argv) > 3 else 0.001, 3, list(range(haps)))
eng = mumemto_amd.Engine(0)
eng.set_docs(docs)
names = ["text", "suffix_sort", "lcp_bwt", "scan_kernel", "verify", "rows", "format", "total"]
for label, kw in [("-k 2", dict(num_distinct=2, max_doc_freq=1)), ("-k %d" % (haps // 2), dict(num_distinct=haps // 2, max_doc_freq=1)),
                  ("-k -10", dict(num_distinct=haps - 10, max_doc_freq=1)), ("-k 2 -f 2", dict(num_distinct=2, max_doc_freq=2))]:
    for rep in range(2):
        t = time.perf_counter(); eng.run(min_match_len=20, **kw); dt = time.perf_counter() - t
    ms = eng.stage_ms()
    print("%-12s %.1f ms  %s  rows %d, candidates %d, output %d bytes" % (label, dt * 1e3,
          {k: round(v, 2) for k, v in zip(names, ms)}, eng.L.mmt_num_rows(eng.h), eng.L.mmt_num_candidates(eng.h), eng.output_size()), flush=True)
