"""GPU box helper: ONE rank's share of BASELINE configs[3] -- {anchor} + group_r of 94 x 3.05 Gbp, strict multi-MUMs with merge
metadata, as one streamed pass -- timed on this GPU (what tests/big_c4.py runs eight times before the fold).
usage: big_share.py [--rank 0] [--ranks 8] [--haps 94] [--length 3050000000] [--div 0.001] [--seed 4] [--reps 1] [--no-checks]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd import dist as mdist
import bigchecks

ap = argparse.ArgumentParser()
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--no-checks", action="store_true")
A = ap.parse_args()
mine = mdist.partition_docs(A.haps, A.ranks)[A.rank]
L0 = A.length
t0 = time.time()
bases, lens = synth.collection_sparse(A.haps, L0, A.div, A.seed, which=mine)
print(json.dumps(dict(generated_s=round(time.time() - t0, 1), docs=len(mine), text_chars=int(2 * len(mine) * (L0 + 1)))), flush=True)
os.environ["MMT_GUIDED_STATS"] = "1"
eng = mumemto_amd.Engine(0)
for rep in range(A.reps):
    t = time.time()
    parts = eng.run_partitioned(None, flat=(bases, lens), merge_metadata=True)
    dt = time.time() - t
    mem = eng.device_memory()
    print(json.dumps(dict(rank=A.rank, ranks=A.ranks, seconds=round(dt, 2), partitions=parts, producer=eng.producer_used(),
                          wide=bool(eng.is_wide()), stage_ms=[round(x) for x in eng.stage_ms()], stream=eng.stream_stats(),
                          memory_gb={k: round(v / 2**30, 1) for k, v in mem.items() if k != "map_seconds"},
                          rows=int(eng.L.mmt_num_rows(eng.h)))), flush=True)
if not A.no_checks:
    bigchecks.check_mum_rows(eng, bases, lens, use_text=False)
print("OK")
