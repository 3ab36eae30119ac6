"""GPU box helper: ONE rank's share of BASELINE configs[3] / configs[4] with REAL-ASSEMBLY-LIKE content at whole-genome size --
`synth.haplotypes_realistic`: satellite arrays of a 171-base monomer (2.3 % + 3.9 % of the length: 70 + 119 Mbp at 3.05 Gbp), twenty
microsatellites, three assembly gaps (2.4 / 9.5 / 48 Mbp) broken into pieces by indels (1e-4 per base), an inversion in every
seventh haplotype, haplotypes of unequal lengths.  The content the reference has no limits for (include/newscan.hpp:265-307 cuts
phrases of any length, include/pfp.hpp:210-244 works on any content) and every full-size share before round 6 did not have.

  --mode strict : {anchor} + group_r of 94, strict multi-MUMs with merge metadata (what a rank of configs[3] runs)
  --mode c5     : the first --haps haplotypes, -k -1 -f 3, the rank's share of the stream only (what a rank of configs[4] runs)

Checks: sampled rows against the definition (bigchecks.check_mum_rows / check_mem_rows); PRECISION AND RECALL inside whole bins
(bigchecks.check_bins_complete) for k-mers INSIDE the satellite arrays, AT the ends of the gap pieces ("NN" + the bases behind a
gap) and at random places; and the instrument's own loop closed: the positions of every bin are enumerated a second time on the
HOST (bytes.find over the haplotypes and their reverse complements, no GPU involved) and must equal what `k_kmer_positions` listed.

usage: big_share_real.py [--mode strict] [--rank 0] [--ranks 8] [--haps 94] [--length 3050000000] [--div 0.001] [--seed 4]
                         [--procs 6] [--iid-seconds S]   (S: the i.i.d. share's time on this code, for the ratio in the report)"""
import argparse, json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from mumemto_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="strict", choices=["strict", "c5"])
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--procs", type=int, default=6)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--iid-seconds", type=float, default=0.0)
ap.add_argument("--no-checks", action="store_true")
A = ap.parse_args()
SHM = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"


def _one(h):
    for _, s in synth.haplotypes_realistic(A.haps, A.length, A.div, A.seed, which=[h]):
        np.save(os.path.join(SHM, "mmt_real_%03d.npy" % h), s)
        return h, len(s)


def _find_all(args):
    """every start of `pat` in the document (bytes.find: memmem speed) -- the HOST's own enumeration of a bin"""
    path, pat = args
    b = np.load(path, mmap_mode="r")
    buf = memoryview(b)
    out, at = [], 0
    data = bytes(buf) if len(b) < (1 << 31) else None
    if data is not None:
        while True:
            at = data.find(pat, at)
            if at < 0:
                break
            out.append(at); at += 1
        return out
    step = 1 << 30                                            # pieces of 1 GB overlapping by the pattern
    for lo in range(0, len(b), step):
        piece = bytes(buf[lo:min(len(b), lo + step + len(pat) - 1)])
        at = 0
        while True:
            at = piece.find(pat, at)
            if at < 0:
                break
            out.append(lo + at); at += 1
    return out


def main():
    from mumemto_amd import dist as mdist
    if A.mode == "strict":
        mine = mdist.partition_docs(A.haps, A.ranks)[A.rank]
    else:
        mine = list(range(A.haps))
    avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
    need_gb = len(mine) * A.length / 2**30 * 2.2 + A.procs * 6.5 * A.length / 2**30 + 16
    print(json.dumps(dict(host_available_gb=round(avail_gb), host_needed_gb=round(need_gb), docs=len(mine))), flush=True)
    if avail_gb < need_gb:
        print("SKIPPED: not enough host memory for the collection")
        sys.exit(3)
    t0 = time.time()
    with mp.Pool(A.procs) as pool:
        got = dict(pool.map(_one, mine, chunksize=1))
    lens = np.array([got[h] for h in mine], np.uint64)
    paths = [os.path.join(SHM, "mmt_real_%03d.npy" % h) for h in mine]
    bases = np.empty(int(lens.sum()), np.uint8)
    at = 0
    for p, l in zip(paths, lens):
        bases[at:at + int(l)] = np.load(p, mmap_mode="r")
        at += int(l)
    anc, feats = synth.realistic_ancestor(A.length, A.seed)
    n_text = int(sum(2 * (int(l) + 1) for l in lens))
    print(json.dumps(dict(generated_s=round(time.time() - t0, 1), docs=len(mine), text_chars=n_text,
                          lengths=[int(l) for l in lens[:4]] + ["..."],
                          features={k: sum(e - s for kk, s, e in feats if kk == k) for k in sorted(set(f[0] for f in feats))})), flush=True)
    import mumemto_amd
    import bigchecks

    # ---- the bins: inside the satellite arrays (k-mers over a doubly diverged copy: rare), behind the assembly gaps ("NN" + the
    # bases behind the run), at random places -- all spelled from the ANCESTOR (any k-mer is a legitimate bin; most haplotypes
    # carry the ancestor's)
    rng = np.random.default_rng(77 + A.rank)
    k = 14
    kmers = []
    # (a k-mer of a periodic region may begin millions of suffixes: every candidate is counted in ALL the periodic regions of the
    # ancestor -- both satellite arrays share their monomer -- and taken when it is rare there)
    periodic = b"|".join(bytes(anc[s:e]) for kind, s, e in feats if kind != "gap")

    def rare(km, limit=40):
        c, at = 0, 0
        while c <= limit:
            at = periodic.find(km, at)
            if at < 0:
                break
            c += 1; at += 1
        return c <= limit
    for kind, s, e in feats:
        if kind != "satellite":
            continue
        found = 0
        for _ in range(3000):
            p = int(rng.integers(s, e - k))
            km = bytes(anc[p:p + k])
            if km in kmers or not rare(km):
                continue
            kmers.append(km); found += 1
            if found == 3:
                break
    n_sat = len(kmers)
    n_gap = 0
    for kind, s, e in feats:
        if kind == "gap" and e + k < len(anc):
            kmers.append(b"NN" + bytes(anc[e:e + k - 2])); n_gap += 1
            kmers.append(bytes(anc[e:e + k])); n_gap += 1              # the first bases behind the gap
    while len(kmers) < n_sat + n_gap + 6:
        p = int(rng.integers(0, len(anc) - k))
        km = bytes(anc[p:p + k])
        if b"N" not in km and km not in kmers and not any(s - k < p < e for _, s, e in feats) and rare(km, 0):
            kmers.append(km)
    del periodic
    del anc
    print(json.dumps(dict(bins=len(kmers), inside_satellites=n_sat, behind_gaps=n_gap, k=k)), flush=True)

    os.environ["MMT_GUIDED_STATS"] = "1"
    eng = mumemto_amd.Engine(0)
    kw = dict(merge_metadata=True) if A.mode == "strict" else dict(num_distinct=len(mine) - 1, max_doc_freq=3)
    if A.mode == "c5":
        eng.set_scan_shard(A.rank, A.ranks)
    if not A.no_checks:
        eng.set_row_tap(kmers, max_rows=1 << 16, max_occ=1 << 22)
    best = None
    for rep in range(A.reps):
        t = time.time()
        parts = eng.run_partitioned(None, flat=(bases, lens), **kw)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
        mem = eng.device_memory()
        print(json.dumps(dict(mode=A.mode, content="realistic", rank=A.rank, ranks=A.ranks, seconds=round(dt, 2), partitions=parts,
                              producer=eng.producer_used(), expanded=bool(eng.producer_expanded()), wide=bool(eng.is_wide()),
                              stage_ms=[round(x) for x in eng.stage_ms()], stream=eng.stream_stats(),
                              pfp_counts=eng.pfp_counts() if hasattr(eng, "pfp_counts") else None,
                              memory_gb={kk: round(v / 2**30, 1) for kk, v in mem.items() if kk != "map_seconds"},
                              rows=int(eng.L.mmt_num_rows(eng.h)),
                              ratio_to_the_iid_share=round(dt / A.iid_seconds, 2) if A.iid_seconds else None)), flush=True)
    assert parts == 1, "the share did not run as one streamed pass"
    if A.no_checks:
        print("OK")
        return
    text = bigchecks.LazyText(bases, lens)
    t = time.time()
    if A.mode == "c5":
        # (a rank of a sharded run holds whole bins of leading characters: the bins of this one's share are the ones to check)
        all_kmers = kmers
        kmers = [km for km in all_kmers if eng.kmer_in_share(km)]
        print(json.dumps(dict(bins_in_this_ranks_share=len(kmers), of=len(all_kmers))), flush=True)
        assert kmers, "none of the bins lies in the rank's share: another --rank or --seed"
    bins, suffixes, rows = bigchecks.check_bins_complete(eng, text, text.n, text.doc_start, kmers, **(
        dict() if A.mode == "strict" else dict(num_distinct=len(mine) - 1, max_doc_freq=3)))
    print(json.dumps(dict(bins_checked=bins, suffixes_sorted_on_the_host=suffixes, rows_in_those_bins_equal_to_the_oracles=rows,
                          check_s=round(time.time() - t, 1))), flush=True)
    # ---- the instrument's own loop: the host enumerates the positions of every bin by itself
    t = time.time()
    pos, which = eng.kmer_positions(kmers, cap=1 << 24)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    # (one bin of every kind: a whole-genome document is seconds of memmem per pattern and strand)
    chosen = sorted({0, n_sat, n_sat + 1, n_sat + n_gap} & set(range(len(kmers)))) if A.mode != "c5" else list(range(min(3, len(kmers))))
    jobs = []
    for i in chosen:
        km = kmers[i]
        for p in paths:
            jobs.append((p, km)); jobs.append((p, km.translate(comp)[::-1]))
    with mp.Pool(min(16, A.procs * 2)) as pool:
        hits = pool.map(_find_all, jobs, chunksize=1)
    j = 0
    for i in chosen:
        km = kmers[i]
        want = []
        for d in range(len(paths)):
            Ld, ds = int(lens[d]), int(text.doc_start[d])
            want += [ds + q for q in hits[j]]; j += 1
            want += [ds + Ld + 1 + (Ld - q - len(km)) for q in hits[j]]; j += 1
        have = np.sort(pos[which == i].astype(np.int64))
        assert np.array_equal(have, np.sort(np.array(want, np.int64))), ("bin %r: the device listed %d positions, the host finds %d"
                                                                           % (km, len(have), len(want)))
    print(json.dumps(dict(bins_enumerated_on_the_host=len(chosen), positions_listed_by_the_device=int(len(pos)), equal_to_k_kmer_positions=True,
                          enumerate_s=round(time.time() - t, 1))), flush=True)
    eng.set_row_tap([])
    if A.mode == "strict":
        bigchecks.check_mum_rows(eng, bases, lens, use_text=False)
    else:
        bigchecks.check_mem_rows(eng, bases, lens, min_docs=len(mine) - 1, max_doc_freq=3, text=text)
    for p in paths:
        os.unlink(p)
    print("OK")


if __name__ == "__main__":
    main()
