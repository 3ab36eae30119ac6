"""GPU box helper: a collection far beyond what one stored suffix array could hold, as ONE streamed run on one GPU.
usage: python tests/big_waves.py HAPS LENGTH [DIVERGENCE] [--grouping FRAC] [--mode strict|partial]
Prints seconds, stage times, producer, batches, device memory; checks sampled rows (tests/bigchecks.py) and, with
--grouping, that anchor partitions + merge give the same bytes (up to the reference's end-of-stream quirk)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
import bigchecks

args = [a for a in sys.argv[1:] if not a.startswith("--")]
haps, length = int(args[0]), int(args[1])
div = float(args[2]) if len(args) > 2 else 0.001
grouping = float(sys.argv[sys.argv.index("--grouping") + 1]) if "--grouping" in sys.argv else 0.0
mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "strict"
t0 = time.time()
bases = np.empty(haps * length, np.uint8)
for h, b in synth.haplotypes_sparse(haps, length, div, 4):
    bases[h * length:(h + 1) * length] = b
lens = np.full(haps, length, np.uint64)
print("generated %d x %d in %.1f s" % (haps, length, time.time() - t0), flush=True)
eng = mumemto_amd.Engine(0)
kw = {} if mode == "strict" else dict(num_distinct=haps - 1, max_doc_freq=3)
os.environ["MMT_GUIDED_STATS"] = "1"
t = time.time()
parts = eng.run_partitioned(None, flat=(bases, lens), **kw)
dt = time.time() - t
n = eng.text_length()
res = dict(haps=haps, length=length, text_chars=n, seconds=round(dt, 2), partitions=parts, producer=eng.producer_used(),
           gbp_per_s=round(haps * length / dt / 1e9, 3), stage_ms=[round(x) for x in eng.stage_ms()], stream=eng.stream_stats(),
           memory_gb={k: round(v / 2**30, 1) for k, v in eng.device_memory().items() if k != "map_seconds"},
           rows=eng.output_text().count(b"\n"), mode=mode)
print(json.dumps(res), flush=True)
single = eng.output_text()
if mode == "strict":
    bigchecks.check_mum_rows(eng, bases, lens)
else:
    bigchecks.check_mem_rows(eng, bases, lens, min_docs=haps - 1, max_doc_freq=3)
if grouping > 0 and mode == "strict":
    os.environ["MMT_MAX_TEXT"] = str(int(n * grouping))
    t = time.time()
    parts = eng.run_partitioned(None, flat=(bases, lens))
    dt2 = time.time() - t
    del os.environ["MMT_MAX_TEXT"]
    part = eng.output_text()
    a, b = set(single.split(b"\n")), set(part.split(b"\n"))
    same = single == part or (len(b - a) == 0 and len(a - b) <= parts)
    print(json.dumps(dict(grouping_partitions=parts, seconds=round(dt2, 2), same_as_one_run=bool(same),
                          rows_missing_by_the_stream_end_quirk=len(a - b))), flush=True)
    assert same
