"""GPU box helper: what ONE rank of `bench.py --gpus N` does on the C3 stand-in -- the anchor + its share of the other 93
haplotypes with merge metadata -- timed on this GPU.  usage: rank_share_probe.py <N>"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd import dist as mdist

N = int(sys.argv[1]); haps, length = 94, 64_000_000
mine = mdist.partition_docs(haps, N)[0]
TAIL = len(sys.argv) > 2 and sys.argv[2] == "tail"      # N = 2 only: also what rank 0 does after the exchange
d = tempfile.mkdtemp(prefix="share_", dir="/dev/shm")
try:
    paths = []
    for h, b in synth.haplotypes_sparse(haps, length, 0.001, 3, which=set(mine)):
        p = os.path.join(d, "h%03d.fa" % h); synth.write_fasta_fast(p, b, name="h%03d" % h); paths.append(p)
    eng = mumemto_amd.Engine(0)
    for rep in range(3):
        t = time.perf_counter()
        sec = eng.run_files(paths, out_prefix=None, merge_metadata=True)
        print("N=%d share of %d haplotypes: %.3f s %s stage ms %s wide %s rows %d cand %d" % (
            N, len(mine), time.perf_counter() - t, {k: round(v, 3) for k, v in sec.items()}, [round(x, 1) for x in eng.stage_ms()],
            eng.is_wide(), eng.L.mmt_num_rows(eng.h), eng.L.mmt_num_candidates(eng.h)), flush=True)
    if TAIL and N == 2:
        part0 = eng.rows_mum() + (eng.thresholds()[: length + 1].copy(),)
        part0 = tuple(np.array(x) for x in part0)
        other = mdist.partition_docs(haps, N)[1]
        paths2 = [paths[0]]
        for h, b in synth.haplotypes_sparse(haps, length, 0.001, 3, which=set(other) - {0}):
            p = os.path.join(d, "h%03d.fa" % h); synth.write_fasta_fast(p, b, name="h%03d" % h); paths2.append(p)
        eng.run_files(paths2, out_prefix=None, merge_metadata=True)
        part1 = tuple(np.array(x) for x in eng.rows_mum() + (eng.thresholds()[: length + 1].copy(),))
        for rep in range(3):
            t = time.perf_counter()
            m = eng.anchor_merge([part0, part1], sort_like_direct=True, want_rows=False, text_file=os.path.join(d, "out.mums"))
            print("rank 0 after the exchange (upload of both partitions' tables, fold, re-sort, format, write): %.3f s, %d rows, %d bytes"
                  % (time.perf_counter() - t, m["n_rows"], os.path.getsize(os.path.join(d, "out.mums"))), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
