"""GPU box helper: head-room.  A rank's share of BASELINE configs[3] -- {anchor + 12} x 3.05 Gbp, 79.3 G text characters, strict
multi-MUMs with merge metadata -- peaks at ~228 GB of the device's 288.  Here the same share runs twice in one process: as it
comes, and with 88 GB of the device declared off limits (mmt_pool_set_reserve: what MUMEMTO_HEAP_RESERVE does for a whole
process), i.e. on a 200 GB device.  The estimate then refuses the text as one suffix array and the engine falls back to anchor
partitions inside the rank + its own fold + re-sort (partitioned.cpp): the rows must be the first run's up to the stream-end
quirk (the reference never closes the last interval of a run, pfp_lcp_mum.hpp:223-230: at most one row per partition is
missing), in the same order, and the heap must have stayed below 200 GB.

usage: big_reserve.py [--share 1] [--reserve-gb 88] [--haps 94] [--length 3050000000] [--ranks 8]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd import dist as mdist

ap = argparse.ArgumentParser()
ap.add_argument("--share", type=int, default=1)
ap.add_argument("--reserve-gb", type=float, default=88.0)
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
A = ap.parse_args()


def row_hashes(L, off, st):
    """one 64-bit value per row (length, every offset, every strand): equal rows give equal values"""
    h = L.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    for d in range(off.shape[1]):
        h ^= (off[:, d].astype(np.uint64) + np.uint64(d + 1)) * np.uint64(0xC2B2AE3D27D4EB4F + 2 * d)
        h = (h << np.uint64(13)) | (h >> np.uint64(51))
        h += st[:, d].astype(np.uint64) * np.uint64(0x165667B19E3779F9)
    return h


def main():
    mine = mdist.partition_docs(A.haps, A.ranks)[A.share]
    L0 = A.length
    t0 = time.time()
    bases, lens = synth.collection_sparse(A.haps, L0, A.div, A.seed, which=mine)
    print(json.dumps(dict(generated_s=round(time.time() - t0, 1), docs=len(mine), text_chars=int(2 * len(mine) * (L0 + 1)))), flush=True)
    lib = mumemto_amd.load_library()
    out = []
    for label, reserve in (("whole device", None), ("%.0f GB off limits" % A.reserve_gb, int(A.reserve_gb * 2**30))):
        lib.mmt_pool_trim()                 # (nothing of the run before is alive: its engine is closed -- the heap starts from nothing)
        lib.mmt_pool_set_reserve(reserve if reserve is not None else 2**64 - 1)
        eng = mumemto_amd.Engine(0)
        base_peak = eng.device_memory()["mapped"]
        t0 = time.time()
        used = eng.run_partitioned(None, flat=(bases, lens), merge_metadata=True)
        dt = time.time() - t0
        L, off, st = eng.rows_mum()
        th = eng.thresholds32()[: L0 + 1].copy()
        mem = eng.device_memory()
        rec = dict(run=label, seconds=round(dt, 1), partitions_inside_the_rank=int(used), producer=eng.producer_used(), rows=int(len(L)),
                   mapped_gb=round(mem["mapped"] / 2**30, 1), peak_gb_of_the_process_so_far=round(mem["peak"] / 2**30, 1),
                   mapped_before_gb=round(base_peak / 2**30, 1))
        print(json.dumps(rec), flush=True)
        out.append((used, L.copy(), off.copy(), st.copy(), th, mem))
        eng.close()
    lib.mmt_pool_set_reserve(2**64 - 1)
    (u1, L1, o1, s1, t1, m1), (u2, L2, o2, s2, t2, m2) = out
    assert u1 == 1 and u2 >= 2, (u1, u2)
    assert m2["mapped"] <= (288 - A.reserve_gb + 2) * 2**30, m2
    h1, h2 = row_hashes(L1, o1, s1), row_hashes(L2, o2, s2)
    missing = np.setdiff1d(h1, h2, assume_unique=False)
    extra = np.setdiff1d(h2, h1, assume_unique=False)
    # the common rows in the same order
    keep1 = np.isin(h1, h2)
    same_order = bool(np.array_equal(h1[keep1], h2[np.isin(h2, h1)]))
    th_equal = float(np.mean(t1 == t2))
    rec = dict(rows_whole=int(len(h1)), rows_partitioned=int(len(h2)), missing_by_the_stream_end_quirk=int(len(missing)),
               rows_not_in_the_whole_run=int(len(extra)), same_order=same_order, thresholds_equal_fraction=round(th_equal, 6),
               thresholds_differ_at=int(np.count_nonzero(t1 != t2)))
    print(json.dumps(rec), flush=True)
    assert len(extra) == 0 and len(missing) <= u2 and same_order
    print("OK")


main()
