import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
def live(tag):
    e = mumemto_amd.Engine(0)
    print(tag, {k: v for k, v in e.device_memory().items()}, flush=True)
    e.close()
live("start")
docs = synth.pangenome(9, 30000, 0.01, seed=4)
for prod, env in (("pfp", {}), ("guided", {"MMT_GUIDED_BATCH": "20000"}), ("expand", {"MMT_GUIDED_BATCH": "6000"})):
    os.environ.update(env)
    eng = mumemto_amd.Engine(0)
    eng.set_producer(prod, 6, 16)
    eng.set_docs(docs)
    eng.run(merge_metadata=True)
    L, off, st = eng.rows_mum()
    th = eng.thresholds32().copy()
    eng.release_columns(keep_anchor_ranks=True)
    m = eng.anchor_merge([(L, off, st, th), (L, off, st, th)], sort_like_direct=True, want_rows=True, slices=4, want_text=False)
    eng.close()
    for k in env: del os.environ[k]
    live("after " + prod)
eng = mumemto_amd.Engine(0)
eng.set_producer("expand", 6, 16)
eng.set_row_tap([b"ACGTACG"])
eng.set_docs(docs); eng.run(); eng.kmer_positions([b"ACGTACG"]); eng.set_row_tap([]); eng.close()
live("after tap")
