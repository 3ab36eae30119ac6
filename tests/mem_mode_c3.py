"""GPU box helper: the scan instantiation of BASELINE configs[4] -- partial multi-MEMs `-k -1 -f 3` (num_distinct N - 1, at most 3
per document, 3 N in all: the non-exact k_scan with a window of N - 2 entries and walks of up to 3 N) -- on the C3 stand-in
(94 x 64 Mbp, 12.03 G characters), two passes.  usage: mem_mode_c3.py [haps] [length] [passes]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth
haps = int(sys.argv[1]) if len(sys.argv) > 1 else 94
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bases = np.empty(haps * L, np.uint8)
for h, b in synth.haplotypes_sparse(haps, L, 0.001, 3):
    bases[h * L:(h + 1) * L] = b
lens = np.full(haps, L, np.uint64)
eng = mumemto_amd.Engine(0)
for rep in range(passes):
    t = time.perf_counter()
    assert eng.run_partitioned(None, flat=(bases, lens), num_distinct=haps - 1, max_doc_freq=3, max_total_freq=3 * haps) == 1
    dt = time.perf_counter() - t
    ms = eng.stage_ms()
    n = eng.text_length()
    print(json.dumps(dict(mode="-k -1 -f 3", haps=haps, length=L, text_chars=n, seconds=round(dt, 3), scan_kernel_ms=round(ms[3], 3),
                          scan_launches=eng.scan_ranges(), rows=int(eng.L.mmt_num_rows(eng.h)), candidates=int(eng.L.mmt_num_candidates(eng.h)),
                          bytes_per_suffix=sum(eng.column_bytes()),
                          roofline_frac=round(sum(eng.column_bytes()) * n / (ms[3] * 1e-3) / 8e12, 4))), flush=True)
