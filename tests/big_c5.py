"""GPU box helper: BASELINE configs[4] in the shape one rank of eight sees it -- partial multi-MEMs (-k -1 -f 3) over a collection
of whole-genome haplotypes that EVERY rank holds in full, the text packed to two bits per character (textref.hpp), the rank
producing, scanning and dropping only its share of the stream (whole bins of leading characters: guided.cpp).

usage: big_c5.py [--haps 41] [--length 3050000000] [--rank 3] [--ranks 8] [--div 0.001] [--seed 4]
Prints seconds, stage times, device memory (peak), the share of the stream the rank produced, rows; checks sampled rows
(every occurrence spells the same string, at least N - 1 documents, at most 3 per document, maximal)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
import bigchecks

ap = argparse.ArgumentParser()
ap.add_argument("--haps", type=int, default=41)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--samples", type=int, default=200)
ap.add_argument("--wp", type=int, nargs=2, default=None, help="window and modulus of the parse (default: automatic)")
ap.add_argument("--supplied", default="auto", choices=["auto", "yes", "no"],
                help="the documents supplied to the engine one at a time (mmt_engine_run_supplied) instead of resident on the host")
ap.add_argument("--out", default="", help="PREFIX: the rank's rows go to PREFIX.mems window by window (a text that fills the device "
                                         "keeps nothing else of them) and the checks read the file")
A = ap.parse_args()
N, L0 = A.haps, A.length
# (the collection sits in host memory as bytes: a box with less memory than that is not asked to try.  What counts is the
# smaller of the machine's MemAvailable and what the container's memory cgroup still allows: a PREFIX on a tmpfs counts
# against both, and a box that runs out of either is lost with everything on it)
def _first_int(paths):
    for q in paths:
        try:
            v = open(q).read().strip()
            if v and v != "max":
                return int(v)
        except OSError:
            pass
    return None


avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
cg_max = _first_int(["/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"])
cg_now = _first_int(["/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"]) or 0
if cg_max is not None and cg_max < (1 << 60):
    avail_gb = min(avail_gb, (cg_max - cg_now) / 2**30)
# the rank's PREFIX.mems: ~25 bytes per occurrence, 0.055 occurrences per suffix of the share (measured at 250 G characters,
# divergence 0.001: 1.67 G occurrences in 30.3 G suffixes), and a third on top
out_gb = 25.0 * 0.055 * (2.0 * N * L0 / A.ranks) / 2**30 * 1.33 if A.out else 0.0
out_fs = os.statvfs(os.path.dirname(os.path.abspath(A.out)) or "/") if A.out else None
out_free_gb = out_fs.f_bavail * out_fs.f_frsize / 2**30 if A.out else 0.0
out_in_memory = bool(A.out) and os.path.abspath(A.out).startswith(("/dev/shm", "/run"))
resident_gb = N * L0 / 2**30 * 1.15 + 16                    # the collection as bytes on the host
supplied_gb = 4.0 * L0 / 2**30 + 0.03 * N + 16              # ancestor, two page-locked documents, substitution lists
supplied = A.supplied == "yes" or (A.supplied == "auto" and avail_gb < resident_gb + (out_gb if out_in_memory else 0.0))
need_gb = (supplied_gb if supplied else resident_gb) + (out_gb if out_in_memory else 0.0)
print(json.dumps(dict(host_available_gb=round(avail_gb), host_needed_gb=round(need_gb), memory_cgroup_max_gb=cg_max and round(cg_max / 2**30),
                      documents="supplied one at a time" if supplied else "resident on the host", output_estimate_gb=round(out_gb),
                      output_dir_free_gb=round(out_free_gb), output_in_memory=out_in_memory)), flush=True)
if avail_gb < need_gb or (A.out and out_free_gb < out_gb):
    print("SKIPPED: not enough host memory (or room for the output) for the collection")
    sys.exit(3)
t0 = time.time()
lens = np.full(N, L0, np.uint64)
model = bigchecks.SparseModel(94, L0, A.div, A.seed, which=list(range(N)))
if supplied:
    bases = model                                    # (the checks read the documents from the model)
else:
    bases = np.empty(N * L0, np.uint8)
    for k in range(N):
        model.fill(k, bases[k * L0:(k + 1) * L0])
n_text = 2 * N * (L0 + 1)
print(json.dumps(dict(generated_s=round(time.time() - t0, 1), haps=N, length=L0, text_chars=n_text)), flush=True)
os.environ["MMT_GUIDED_STATS"] = "1"
eng = mumemto_amd.Engine(0)
eng.set_scan_shard(A.rank, A.ranks)
if A.wp:
    eng.set_producer("guided", A.wp[0], A.wp[1])
if A.out:
    eng.set_text_sink(A.out + ".mems")
t = time.time()
if supplied:
    t_fill = [0.0]

    def supplier(d, dst):
        t1 = time.time()
        model.fill(d, dst)
        t_fill[0] += time.time() - t1
    parts = eng.run_supplied(lens, supplier, num_distinct=N - 1, max_doc_freq=3)
    print(json.dumps(dict(supplier_s=round(t_fill[0], 1))), flush=True)
else:
    parts = eng.run_partitioned(None, flat=(bases, lens), num_distinct=N - 1, max_doc_freq=3)
dt = time.time() - t
eng.set_text_sink(None)
mem = eng.device_memory()
pieces = eng.sort_pieces()
st = eng.stream_stats()
kept = not A.out or not os.path.exists(A.out + ".mems") or os.path.getsize(A.out + ".mems") == 0 or eng.L.mmt_num_occ(eng.h) > 0
L, occ, off, ids, strands = eng.rows_mem() if kept else (np.zeros(eng.L.mmt_num_rows(eng.h), np.uint8), None, [], None, None)
print(json.dumps(dict(mode="-k -1 -f 3", rank=A.rank, ranks=A.ranks, text_chars=eng.text_length(), seconds=round(dt, 1),
                      one_run=parts == 1, producer=eng.producer_used(), wide=bool(eng.is_wide()),
                      stage_ms=[round(x) for x in eng.stage_ms()],
                      share_of_the_stream=dict(first_entry=pieces[A.rank][0], entries=pieces[A.rank][1],
                                               fraction=round(pieces[A.rank][1] / eng.text_length(), 4), produced=st["entries"],
                                               windows=st["windows"], window_bytes=st["window_bytes"]),
                      memory_gb={k: round(v / 2**30, 1) for k, v in mem.items() if k != "map_seconds"}, heap_map_seconds=round(mem.get("map_seconds", 0.0), 2),
                      rows=int(len(L)), occurrences=int(len(off)) if kept else None,
                      output_bytes=os.path.getsize(A.out + ".mems") if A.out else None, rows_kept_on_the_device=bool(kept))), flush=True)
assert parts == 1 and eng.producer_used() == "guided" and (eng.is_wide() or n_text < 2**32)
assert st["entries"] == pieces[A.rank][1], "the rank produced something else than its share of the stream"
assert abs(pieces[A.rank][1] / eng.text_length() - 1.0 / A.ranks) < 0.05
if kept:
    bigchecks.check_mem_rows(eng, bases, lens, min_docs=N - 1, max_doc_freq=3, samples=A.samples, text=bigchecks.LazyText(bases, lens))
else:
    bigchecks.check_mems_file(A.out + ".mems", bigchecks.LazyText(bases, lens), lens, min_docs=N - 1, max_doc_freq=3, samples=A.samples)
if A.out:
    if os.path.getsize(A.out + ".mems") < (8 << 30):          # (small runs: the bytes of two ways of cutting the batches are compared)
        import hashlib
        print(json.dumps(dict(output_sha256=hashlib.sha256(open(A.out + ".mems", "rb").read()).hexdigest())), flush=True)
    os.unlink(A.out + ".mems")
print("OK")
