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
    dev = torch.device("cuda", 0)
    t = time.perf_counter()
    lt, ot, stt = mdist.engine_rows_as_tensors(eng, dev)
    th = torch.as_tensor(mdist.DevicePointerView(eng.thresh_device_ptr(), L + 1), device=dev)
    parts.append((lt.clone(), ot.clone(), stt.clone(), th.clone()))
    torch.cuda.synchronize()
    print("rank", r, "run %.1f ms rows %d, views+clones %.2f ms" % (dt * 1e3, len(lt), (time.perf_counter() - t) * 1e3), flush=True)
parts.reverse()
os.environ['MMT_MERGE_DEBUG']='1'
for rep in range(2):
    t = time.perf_counter()
    m = eng.anchor_merge(mdist.device_partitions(parts), sort_like_direct=True, want_rows=False)
    print("merge of %d partitions: %.1f ms, %d rows x %d docs, %d bytes" % (G, (time.perf_counter() - t) * 1e3, m["n_rows"], m["n_docs"], len(m["text"])))
os.environ.pop('MMT_MERGE_DEBUG')
for rep in range(3):
    t = time.perf_counter()
    m = eng.anchor_merge(mdist.device_partitions(parts), sort_like_direct=True, want_rows=False)
    print("  without debug syncs: %.1f ms" % ((time.perf_counter() - t) * 1e3))
