"""GPU box helper: time the rank-0 merge of the multi-GPU path for G emulated ranks."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mumemto_amd
from mumemto_amd import synth, dist as mdist

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12_100_000
haps = 1 + 15 * G
groups = mdist.partition_docs(haps, G)
eng = mumemto_amd.Engine(0)
parts = []
for r in reversed(range(G)):
    docs = synth.pangenome_subset(haps, L, 0.005, 2, groups[r])
    eng.set_docs(docs)
    t = time.perf_counter(); eng.run(merge_metadata=True); dt = time.perf_counter() - t
    length, off, st = eng.rows_mum()
    o = np.argsort(off[:, 0], kind='stable'); length, off, st = length[o], off[o], st[o]
    parts.append((length, off, st, eng.thresholds()[: L + 1].copy()))
    print("rank", r, "run %.1f ms rows %d" % (dt * 1e3, len(length)), flush=True)
parts.reverse()
os.environ['MMT_MERGE_DEBUG']='1'
for rep in range(2):
    t = time.perf_counter()
    m = eng.anchor_merge(parts, sort_like_direct=True)
    print("merge of %d partitions: %.1f ms, %d rows x %d docs, %d bytes" % (G, (time.perf_counter() - t) * 1e3, len(m["lengths"]), m["offsets"].shape[1], len(m["text"])))
