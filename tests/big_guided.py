"""GPU box helper: a collection with little redundancy (the anchor next to a few whole haplotypes) through the guided
producer, compared with the prefix-free parse proper when that fits, and checked by size-independent properties.
usage: big_guided.py <haps> <length> [divergence] [compare: pfp|none|auto] [checks: full|light|none]
compare = auto: the automatic producer alone, which must fall back to the guided sort by itself;
compare = parts: the same, for a collection that needs anchor partitions + merge (BASELINE configs[3] in small)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth
import bigchecks

haps, length = int(sys.argv[1]), int(sys.argv[2])
div = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
compare = sys.argv[4] if len(sys.argv) > 4 else "pfp"
checks = sys.argv[5] if len(sys.argv) > 5 else "light"
wp = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 0)      # PFP window / modulus of a forced guided run
t = time.perf_counter()
bases = np.empty(haps * length, np.uint8)
for h, b in synth.haplotypes_sparse(haps, length, div, 11):
    bases[h * length:(h + 1) * length] = b
lens = np.full(haps, length, np.uint64)
n_text = 2 * haps * (length + 1)
print("generated %d x %d bp in %.1f s; text = %.3f G chars" % (haps, length, time.perf_counter() - t, n_text / 1e9), flush=True)
eng = mumemto_amd.Engine(0)
out = {}
passes = []
for kind in (["guided", "pfp"] if compare == "pfp" else ["guided"]):
    eng.set_producer("guided" if kind == "guided" and compare not in ("auto", "parts") else "auto", *wp)
    for rep in range(2 if kind == "guided" else 1):
        t = time.perf_counter()
        parts = eng.run_partitioned(None, flat=(bases, lens))
        dt = time.perf_counter() - t
        passes.append((parts, eng.output_text()))
        print("%s pass %d: %.2f s (%.3f Gbp/s), producer %s, partitions %d, wide %s, rows %d, output %d bytes\n  stage ms %s\n  pfp %s %s\n  memory %s"
              % (kind, rep, dt, haps * length / dt / 1e9, eng.producer_used(), parts, eng.is_wide(), eng.L.mmt_num_rows(eng.h),
                 eng.output_size(), [round(x, 1) for x in eng.stage_ms()], eng.pfp_counts(), [round(x, 1) for x in eng.pfp_stage_ms()],
                 eng.device_memory()), flush=True)
    assert parts == 1 or compare == "parts"
    out[kind] = eng.output_text()
    if kind == "guided":
        assert eng.producer_used() == "guided"
        if checks != "none":
            if parts == 1:
                bigchecks.check_stream(eng, bases, lens, light=(checks == "light"))
            bigchecks.check_mum_rows(eng, bases, lens)
if compare == "pfp":
    print("guided == pfp:", out["guided"] == out["pfp"])
    assert out["guided"] == out["pfp"]
if len({p for p, _ in passes}) > 1:      # the automatic choice took one suffix array once and partitions once (what the heap had left)
    (pa, ta), (pb, tb) = passes[0], passes[-1]
    print("%d partition(s): %d bytes; %d partition(s): %d bytes; identical: %s" % (pa, len(ta), pb, len(tb), ta == tb), flush=True)
    if ta != tb:            # the reference's end-of-stream quirk may drop one row per partition
        a, b = set(ta.split(b"\n")), set(tb.split(b"\n"))
        print("  rows only in the first: %d, only in the last: %d" % (len(a - b), len(b - a)), flush=True)
        assert len(b - a) == 0 and len(a - b) <= max(pa, pb)
print("OK")
