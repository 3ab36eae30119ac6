"""GPU box helper: realistic content at growing size through the producers, with stage times.
usage: python tests/realistic_probe.py HAPS LENGTH [producers...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
haps, length = int(sys.argv[1]), int(sys.argv[2])
kinds = sys.argv[3:] or ["pfp", "guided"]
t0 = time.time()
seqs = [s for _, s in synth.haplotypes_realistic(haps, length, 0.001, 3)]
lens = np.array([len(s) for s in seqs], np.uint64)
bases = np.concatenate(seqs)
del seqs
print("generated %d x ~%d in %.1f s" % (haps, length, time.time() - t0), flush=True)
eng = mumemto_amd.Engine(0)
out = {}
for kind in kinds:
    eng.set_producer(kind) if kind != "auto" else eng.set_producer("auto")
    for rep in range(2):
        t = time.time()
        try:
            parts = eng.run_partitioned(None, flat=(bases, lens))
        except Exception as e:
            print(kind, "FAILED:", str(e)[:300], flush=True)
            break
        dt = time.time() - t
    else:
        out[kind] = eng.output_text()
        print("%-7s %.2f s  partitions %d  producer %s  stages %s  rows %d  mem %s  pfp %s" % (
            kind, dt, parts, eng.producer_used(), ["%.0f" % x for x in eng.stage_ms()], out[kind].count(b"\n"),
            {k: round(v / 2**30, 1) for k, v in eng.device_memory().items() if k != "map_seconds"}, eng.pfp_counts()), flush=True)
ks = list(out)
for k in ks[1:]:
    print(k, "== %s:" % ks[0], out[k] == out[ks[0]], flush=True)
