"""GPU box helper: direct-run thresholds against the fold of partitions, at sizes the oracle cannot reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth

def collection(h, L, d, seed):
    b = np.empty(h * L, np.uint8)
    for i, x in synth.haplotypes_sparse(h, L, d, seed):
        b[i * L:(i + 1) * L] = x
    return b, np.full(h, L, np.uint64)

def direct(eng, bases, lens, wide, rng=None):
    env = {}
    if wide: env["MMT_FORCE_WIDE"] = "1"
    if rng: env["MMT_SCAN_RANGE"] = str(rng)
    os.environ.update(env)
    try:
        d_docs = [[bases[int(i) * int(lens[0]):(int(i) + 1) * int(lens[0])].tobytes()] for i in range(len(lens))]
        eng.set_docs(d_docs)
        eng.run(merge_metadata=True)
        return eng.thresholds()[: int(lens[0]) + 1].copy(), eng.output_text()
    finally:
        for k in env: del os.environ[k]

h, L, d, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
bases, lens = collection(h, L, d, seed)
eng = mumemto_amd.Engine(0)
n_text = 2 * h * (L + 1)
os.environ["MMT_MAX_TEXT"] = str(int(n_text * 0.4))
parts = eng.run_partitioned(None, flat=(bases, lens))
del os.environ["MMT_MAX_TEXT"]
merged = eng.merged_thresholds(L)
mtext = eng.output_text()
for label, wide, rng in (("narrow", False, None), ("narrow ranges", False, 1 << 24), ("wide", True, None)):
    if n_text >= 2 ** 32 - 4096 and not wide:
        continue
    th, text = direct(eng, bases, lens, wide, rng)
    diff = np.nonzero(th != merged)[0]
    print("%-14s partitions %d, rows equal %s, thresholds nonzero %d, differ %d: %s" % (
        label, parts, text == mtext, int((th > 0).sum()), len(diff), [(int(i), int(th[i]), int(merged[i])) for i in diff[:12]]), flush=True)
