"""GPU box helper: BASELINE configs[4] END TO END on one GPU -- 94 whole-genome haplotypes, partial multi-MEMs `-k -1 -f 3`, the
ranks of an 8-GPU run time-multiplexed.  Every rank of that run holds all 573.4 G characters (the reference refuses to merge these
modes from partitions: include/pfp_mum.hpp:178-183) and produces, scans and drops its own share of the stream -- whole bins of
leading characters --; PREFIX.mems is the ranks' pieces in rank order (mumemto_exec --gpus N joins them).  Here rank r = 0 .. 7
run one after the other on the same device: documents supplied one at a time, rows formatted, copied out, digested and dropped
(sink /dev/null: eight pieces of ~66 GB have no place on the box).  Per rank: seconds, suffixes, rows, bytes, digest, peak HBM,
and PRECISION AND RECALL inside whole bins of the rank's share (bigchecks.check_bins_complete against the oracle's scan).
At the end: the shares tile the stream exactly, the sums, the slowest share (the 8-GPU wall clock) and the 1-GPU wall clock.

usage: big_c5_all.py [--haps 94] [--length 3050000000] [--ranks 8] [--only r ...] [--bins 40]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import mumemto_amd
import bigchecks
import pyoracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--only", type=int, nargs="*", default=None)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--bins", type=int, default=40, help="14-mers of the anchor offered to the row tap (those of the rank's share are checked)")
A = ap.parse_args()
N, L0 = A.haps, A.length
avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
if avail_gb < 4.0 * L0 / 2**30 + 0.03 * N + 16:
    print("SKIPPED: not enough host memory for the generator's model")
    sys.exit(3)
t_all = time.time()
model = bigchecks.SparseModel(N, L0, A.div, A.seed, which=list(range(N)))
lens = np.full(N, L0, np.uint64)
n_text = 2 * N * (L0 + 1)
nd, f, mf = O.cli_params(N, k=-1, f=3)
anchor = model.doc(0)[0:min(L0, 4_000_000)]
rng = np.random.default_rng(77)
kmers = []
while len(kmers) < A.bins:
    p = int(rng.integers(0, len(anchor) - 14))
    km = bytes(anchor[p:p + 14])
    if km not in kmers:
        kmers.append(km)
print(json.dumps(dict(model_s=round(time.time() - t_all, 1), haps=N, length=L0, text_chars=n_text, num_distinct=nd, max_doc_freq=f,
                      max_total_freq=mf)), flush=True)
text = bigchecks.LazyText(model, lens)
shares = []
for r in (A.only if A.only else range(A.ranks)):
    mumemto_amd.load_library().mmt_pool_trim()
    eng = mumemto_amd.Engine(0)
    eng.set_scan_shard(r, A.ranks)
    eng.set_text_sink("/dev/null")
    eng.set_row_tap(kmers, max_rows=1 << 16, max_occ=1 << 23)
    t_fill = [0.0]

    def supplier(d, dst):                                 # (the TEST's generator: its time is reported apart from the engine's)
        t1 = time.time()
        model.fill(d, dst)
        t_fill[0] += time.time() - t1
    t0 = time.time()
    parts = eng.run_supplied(lens, supplier, num_distinct=nd, max_doc_freq=f, max_total_freq=mf)
    dt = time.time() - t0
    eng.set_text_sink(None)
    assert parts == 1 and eng.is_wide() and eng.text_length() == n_text and eng.producer_used() == "guided"
    pieces = eng.sort_pieces()
    st = eng.stream_stats()
    assert st["entries"] == pieces[r][1], "the rank produced something else than its share of the stream"
    mem = eng.device_memory()
    written, digest = eng.text_sink_digest()
    rows = int(eng.L.mmt_num_rows(eng.h))
    mine = [km for km in kmers if eng.kmer_in_share(km)]
    t1 = time.time()
    bins, suffixes, tapped = (0, 0, 0)
    check_error = None
    if mine:
        try:
            bins, suffixes, tapped = bigchecks.check_bins_complete(eng, text, text.n, text.doc_start, mine, num_distinct=nd,
                                                                   max_doc_freq=f, max_total_freq=mf)
        except AssertionError as ex:                     # (the other ranks still run; the script fails at the end)
            check_error = str(ex)[:400]
    rec = dict(rank=r, ranks=A.ranks, seconds=round(dt, 1), supplier_s=round(t_fill[0], 1), engine_s=round(dt - t_fill[0], 1),
               expanded=bool(eng.producer_expanded()), first_entry=int(pieces[r][0]), entries=int(pieces[r][1]),
               fraction=round(pieces[r][1] / n_text, 4), windows=st["windows"], rows=rows, bytes=int(written), digest="%016x" % digest,
               peak_hbm_gb=round(mem["peak"] / 2**30, 1), stage_ms=[round(x) for x in eng.stage_ms()],
               bins_checked=bins, suffixes_sorted_on_the_host=suffixes, rows_in_those_bins_equal_to_the_oracles=tapped,
               check_s=round(time.time() - t1, 1), check_error=check_error)
    shares.append(rec)
    print(json.dumps(rec), flush=True)
    eng.set_row_tap([])
    eng.close()
tiles = None
if not A.only:
    at, tiles = 0, True
    for s in shares:
        tiles = tiles and s["first_entry"] == at
        at += s["entries"]
    tiles = bool(tiles and at == n_text)
print(json.dumps(dict(config="configs[4]: %d x %d bp, -k -1 -f 3, %d ranks time-multiplexed on one GPU" % (N, L0, A.ranks),
                      shares=len(shares), shares_tile_the_stream_exactly=tiles, suffixes=sum(s["entries"] for s in shares), rows=sum(s["rows"] for s in shares),
                      bytes=sum(s["bytes"] for s in shares), shares_run_s=round(sum(s["seconds"] for s in shares), 1),
                      slowest_share_s=max(s["seconds"] for s in shares), slowest_share_engine_s=max(s["engine_s"] for s in shares),
                      supplier_s=round(sum(s["supplier_s"] for s in shares), 1), peak_hbm_gb=max(s["peak_hbm_gb"] for s in shares),
                      bins_checked=sum(s["bins_checked"] for s in shares),
                      rows_in_those_bins=sum(s["rows_in_those_bins_equal_to_the_oracles"] for s in shares),
                      input_gbp=round(N * L0 / 1e9, 1), total_s=round(time.time() - t_all, 1))), flush=True)
assert tiles is not False, "the shares do not tile the stream"
assert not any(s["check_error"] for s in shares), "a bin of some rank differs from the oracle's scan"
print("OK")
