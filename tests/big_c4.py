"""GPU box helper: BASELINE configs[3] END TO END on one GPU -- 94 whole-genome haplotypes, strict multi-MUMs, the ranks of an
8-GPU run time-multiplexed: for every rank r the share {anchor} + group_r is generated (never more than one share in host
memory), run as one streamed pass with merge metadata, its rows and its u16[L0 + 1] threshold column kept on the host; then
the G shares are folded by `mmt_anchor_merge_by_ranges` (the fold over anchor coordinate ranges that >= 4 ranks take, one
slice per rank, README.md:124-141 / src/merge_candidates.cpp:106-157,170-255), re-sorted into direct-run order and written
as PREFIX.mums.  Sampled rows of the merged file are checked against the generator's model (every document spells the same
string at its offset, not extendable on either side in all documents at once), the head of the file is in lexicographic
order, and with --verify-direct (collections that also fit as one run) the bytes are compared with the direct run's.

usage: big_c4.py [--haps 94] [--length 3050000000] [--ranks 8] [--slices 0(=ranks)] [--div 0.001] [--seed 4]
                 [--out /dev/shm/c4] [--also-ranks 6] [--verify-direct] [--keep]"""
import argparse, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd import dist as mdist

ap = argparse.ArgumentParser()
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=3_050_000_000)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--slices", type=int, default=0)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=4)
ap.add_argument("--out", default="/dev/shm/c4")
ap.add_argument("--also-ranks", type=int, default=0, help="a second grouping: its bytes must equal the first one's")
ap.add_argument("--verify-direct", action="store_true")
ap.add_argument("--samples", type=int, default=200)
ap.add_argument("--keep", action="store_true")
ap.add_argument("--expect-sha", default="", help="sha256 of PREFIX.mums from another grouping of the same collection")
ap.add_argument("--no-file", action="store_true",
                help="the merged rows are formatted and copied out in pieces as for the file, but written to /dev/null (bench.py's "
                     "whole_genome_1gpu leg: 50 GB of PREFIX.mums on a tmpfs would count against the container's memory); the "
                     "checks of the file are skipped, the merged rows' count and the timings stay")
ap.add_argument("--json-out", default="", help="write the summary record of the first grouping there")
A = ap.parse_args()
N, L0 = A.haps, A.length
_COMP = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


def log(**kw):
    print(json.dumps(kw), flush=True)


def sha_of(path):
    h, n = hashlib.sha256(), 0
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 26)
            if not b:
                break
            h.update(b); n += len(b)
    return h.hexdigest(), n


def run_grouping(eng, G, slices, out_path):
    groups = mdist.partition_docs(N, G)
    parts, t_all = [], time.time()
    per_share = []
    for r, mine in enumerate(groups):
        t0 = time.time()
        bases, lens = synth.collection_sparse(N, L0, A.div, A.seed, which=mine)
        t_gen = time.time() - t0
        t0 = time.time()
        used = eng.run_partitioned(None, flat=(bases, lens), merge_metadata=True)
        t_run = time.time() - t0
        t0 = time.time()
        length, off, st = eng.rows_mum()
        th = eng.thresholds32()[: L0 + 1].copy()            # 32 bits, as the exchange carries them
        parts.append((length.copy(), off.copy(), st.copy(), th))
        mem = eng.device_memory()
        rec = dict(share=r, docs=len(mine), text_chars=int(2 * len(mine) * (L0 + 1)), generate_s=round(t_gen, 1),
                   run_s=round(t_run, 2), copy_out_s=round(time.time() - t0, 1), rows=int(len(length)),
                   partitions_inside_the_rank=used, producer=eng.producer_used(), wide=bool(eng.is_wide()),
                   peak_hbm_gb=round(mem["peak"] / 2**30, 1), stage_ms=[round(x) for x in eng.stage_ms()])
        per_share.append(rec)
        log(**rec)
        del bases
    t0 = time.time()
    eng.release_columns(keep_anchor_ranks=True)       # what a rank holds between its pass and the fold: rows, thresholds, anchor ranks
    m = eng.anchor_merge(parts, sort_like_direct=True, want_rows=False, text_file="/dev/null" if A.no_file else out_path, slices=slices)
    t_fold = time.time() - t0
    sha, size = ("", 0) if A.no_file else sha_of(out_path)
    mem = eng.device_memory()
    rec = dict(grouping=G, slices=slices, shares=per_share, generate_s=round(sum(s["generate_s"] for s in per_share), 1),
               shares_run_s=round(sum(s["run_s"] for s in per_share), 1),
               fold_resort_write_s=round(t_fold, 1), total_s=round(time.time() - t_all, 1), merged_rows=int(m["n_rows"]),
               columns=int(m["n_docs"]), bytes=size, sha256=sha, peak_hbm_gb=round(mem["peak"] / 2**30, 1),
               slowest_share_s=max(s["run_s"] for s in per_share),
               projection_8_gpus_s=round(max(s["run_s"] for s in per_share) + t_fold, 1))
    log(**rec)
    return rec, groups


class Model:
    """The generator's model, sparse: ancestor + per-haplotype substitution lists (synth.haplotypes_sparse)."""

    def __init__(self):
        self.anc = synth.ancestor_codes(A.seed, L0)
        self.ascii = np.frombuffer(b"ACGT", np.uint8)
        self.sub = {}

    def subs(self, h):
        if h not in self.sub:
            hrng = np.random.default_rng([A.seed, h + 1])
            k = int(hrng.binomial(L0, A.div)) if A.div > 0 else 0
            pos = hrng.integers(0, L0, size=k)
            val = (self.anc[pos] + hrng.integers(1, 4, size=k, dtype=np.uint8)) & 3
            # seq[pos] = val: the last write to a position wins
            order = np.argsort(pos, kind="stable")
            pos, val = pos[order], val[order]
            last = np.ones(len(pos), bool)
            last[:-1] = pos[1:] != pos[:-1]
            self.sub[h] = (pos[last], val[last])
        return self.sub[h]

    def piece(self, h, a, b):
        """bases [a, b) of haplotype h, '$' outside the document"""
        out = np.full(b - a, 36, np.uint8)
        lo, hi = max(a, 0), min(b, L0)
        if hi > lo:
            seg = self.anc[lo:hi].copy()
            pos, val = self.subs(h)
            i, j = np.searchsorted(pos, lo), np.searchsorted(pos, hi)
            seg[pos[i:j] - lo] = val[i:j]
            out[lo - a:hi - a] = self.ascii[seg]
        return out


def check_merged_file(path, order, size, samples):
    M = Model()
    rng = np.random.default_rng(7)
    checked = 0
    with open(path, "rb") as f:
        for at in np.sort(rng.integers(0, max(size - 1, 1), size=samples)):
            f.seek(int(at))
            f.readline()
            line = f.readline()
            if not line.endswith(b"\n"):
                continue
            ln, offs, sts = line[:-1].split(b"\t")
            ln = int(ln); offs = [int(x) for x in offs.split(b",")]; sts = sts.split(b",")
            assert len(offs) == N and len(sts) == N, "row does not have one column per document"
            segs, lefts, rights = set(), set(), set()
            for col, h in enumerate(order):
                o = offs[col]
                w = M.piece(h, o - 1, o + ln + 1)
                if sts[col] == b"+":
                    segs.add(w[1:-1].tobytes()); lefts.add(int(w[0])); rights.add(int(w[-1]))
                else:
                    segs.add(_COMP[w[1:-1][::-1]].tobytes()); lefts.add(int(_COMP[w[-1]])); rights.add(int(_COMP[w[0]]))
            assert len(segs) == 1 and len(next(iter(segs))) == ln and ln >= 20, ("row is not a match in every document", line[:80])
            assert len(lefts) > 1 and len(rights) > 1, ("row is not maximal", line[:80])
            checked += 1
        # the head of the file in lexicographic order of the match (the anchor spells it)
        f.seek(0)
        keys = []
        for _ in range(20000):
            line = f.readline()
            if not line:
                break
            ln, offs, _s = line.split(b"\t", 2)
            o = int(offs.split(b",", 1)[0])
            keys.append(M.piece(order[0], o, o + int(ln)).tobytes())
        assert keys == sorted(keys), "merged rows are not in the order of a direct run"
    log(checked_rows=checked, head_rows_in_lexicographic_order=len(keys))


eng = mumemto_amd.Engine(0)
os.environ.setdefault("MMT_MERGE_DEBUG", "1")
slices = A.slices or A.ranks
first, groups = run_grouping(eng, A.ranks, slices, A.out + ".mums")
if A.json_out:
    with open(A.json_out, "w") as f:
        json.dump(first, f)
if A.no_file:
    print("OK")
    sys.exit(0)
check_merged_file(A.out + ".mums", mdist.merged_column_order(groups), first["bytes"], A.samples)
if A.expect_sha:
    log(expected_sha256=A.expect_sha, identical_bytes=first["sha256"] == A.expect_sha)
    assert first["sha256"] == A.expect_sha, "this grouping's bytes differ from the other grouping's"
if A.also_ranks:
    second, _ = run_grouping(eng, A.also_ranks, A.also_ranks, A.out + ".b.mums")
    # the column order of a fold is partition 0's documents, then the others' without the anchor: contiguous blocks give
    # the same order for any number of ranks
    same = second["sha256"] == first["sha256"]
    log(second_grouping=A.also_ranks, identical_bytes=same)
    assert same, "two groupings of the same collection gave different bytes"
    if not A.keep:
        os.unlink(A.out + ".b.mums")
if A.verify_direct:
    bases = np.empty(N * L0, np.uint8)
    for h, b in synth.haplotypes_sparse(N, L0, A.div, A.seed):
        bases[h * L0:(h + 1) * L0] = b
    t0 = time.time()
    used = eng.run_partitioned(None, flat=(bases, np.full(N, L0, np.uint64)))
    direct = eng.output_text()
    with open(A.out + ".mums", "rb") as f:
        merged = f.read()
    a, b = set(direct.split(b"\n")), set(merged.split(b"\n"))
    # the reference never closes the last interval of a run (pfp_lcp_mum.hpp:223-230): a partition whose last interval is
    # a MUM loses that row, at most one row per share
    same = direct == merged or (len(b - a) == 0 and len(a - b) <= A.ranks)
    log(direct_run_s=round(time.time() - t0, 2), partitions=used, identical_to_direct=direct == merged,
        rows_missing_by_the_stream_end_quirk=len(a - b), rows_not_in_direct=len(b - a))
    assert same
if not A.keep:
    os.unlink(A.out + ".mums")
print("OK")
