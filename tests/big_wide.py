"""GPU box helper: a collection whose text exceeds 2^32 characters as ONE suffix array (40-bit positions), checked by
size-independent properties and against the anchor-partition path.
usage: big_wide.py <haps> <length> [divergence] [mode: mum|mem] [checks: full|light|none]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth
import bigchecks

haps, length = int(sys.argv[1]), int(sys.argv[2])
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
mode = sys.argv[4] if len(sys.argv) > 4 else "mum"
checks = sys.argv[5] if len(sys.argv) > 5 else "full"
t = time.perf_counter()
bases = np.empty(haps * length, np.uint8)
for h, b in synth.haplotypes_sparse(haps, length, div, 7):
    bases[h * length:(h + 1) * length] = b
lens = np.full(haps, length, np.uint64)
n_text = 2 * haps * (length + 1)
print("generated %d x %d bp in %.1f s; text = %.3f G chars (2^32 = 4.295 G)" % (haps, length, time.perf_counter() - t, n_text / 1e9), flush=True)
kw = dict(num_distinct=0, max_doc_freq=1, max_total_freq=0) if mode == "mum" else dict(num_distinct=haps - 1, max_doc_freq=3, max_total_freq=0)
eng = mumemto_amd.Engine(0)
for rep in range(2):
    t = time.perf_counter()
    parts = eng.run_partitioned(None, flat=(bases, lens), **kw)
    dt = time.perf_counter() - t
    print("pass %d: %.2f s (%.3f Gbp/s), partitions %d, wide %s, scan ranges %d, rows %d, output %d bytes\n  stage ms %s\n  pfp %s %s\n  memory %s"
          % (rep, dt, haps * length / dt / 1e9, parts, eng.is_wide(), eng.scan_ranges(), eng.L.mmt_num_rows(eng.h), eng.output_size(),
             [round(x, 1) for x in eng.stage_ms()], eng.pfp_counts(), [round(x, 1) for x in eng.pfp_stage_ms()], eng.device_memory()), flush=True)
assert parts == 1 and (eng.is_wide() or n_text < 2 ** 32 - 4096)
single = eng.output_text()
if checks != "none":
    bigchecks.check_stream(eng, bases, lens, light=(checks == "light"))
    if mode == "mum":
        bigchecks.check_mum_rows(eng, bases, lens)
if mode == "mum":
    os.environ["MMT_MAX_TEXT"] = str(int(n_text * 0.4))
    t = time.perf_counter()
    parts = eng.run_partitioned(None, flat=(bases, lens), **kw)
    dt = time.perf_counter() - t
    del os.environ["MMT_MAX_TEXT"]
    part = eng.output_text()
    print("partitioned: %.2f s, %d partitions, output %d bytes, identical to the single suffix array: %s"
          % (dt, parts, len(part), part == single), flush=True)
    if part != single:      # the reference's end-of-stream quirk may drop one row per partition
        a, b = set(single.split(b"\n")), set(part.split(b"\n"))
        print("  rows only in single: %d, only in partitioned: %d" % (len(a - b), len(b - a)))
        assert len(a - b) <= parts and len(b - a) == 0
print("OK")
