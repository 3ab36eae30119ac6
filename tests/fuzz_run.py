"""GPU box helper: the differential test of test_gpu_random.py over many more seeds, plus mid-size collections
(several k_scan tiles, bigger dictionaries).  usage: fuzz_run.py <first seed> <n seeds> [cases per seed]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import numpy as np                                   # noqa: E402
import pyoracle as O                                 # noqa: E402
import mumemto_amd                                   # noqa: E402
from mumemto_amd import synth                        # noqa: E402
from test_gpu_random import random_collection, random_params    # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 40
BIG = len(sys.argv) > 4 and sys.argv[4] == "big"     # every case a pangenome of 0.2 - 3 M text characters
ADV = len(sys.argv) > 4 and sys.argv[4] == "adv"     # mid-size pangenomes with runs, arrays, copies (see adversarial())
SHARD = os.environ.get("MMT_FUZZ_SHARDS") is not None  # also run every case as 2-5 shards of the scan (set_scan_shard)


def adversarial(rng):
    """2-8 haplotypes of 4-30 kbp carrying what makes suffix sorting, LCP construction and the scan work hardest: runs of
    N / of one base up to 12 kbp, tandem arrays, exact copies of another document, large deletions.  Run it with
    MMT_GIANT_RANGE / MMT_SCAN_WIDE_AT / MMT_LONG_CAP lowered so that the rare paths are the common ones."""
    nd_ = int(rng.integers(2, 9))
    docs = synth.pangenome(nd_, int(rng.integers(4000, 30000)), float(rng.choice([0.0, 0.002, 0.01, 0.05])),
                           seed=int(rng.integers(0, 1 << 30)))
    out = []
    for d in range(nd_):
        s = docs[d][0]
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.integers(0, 5))
            a = int(rng.integers(0, len(s)))
            if kind == 0:
                s = s[:a] + b"N" * int(rng.integers(50, 12000)) + s[a:]
            elif kind == 1:
                s = s[:a] + bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(50, 12000)) + s[a:]
            elif kind == 2:
                unit = s[a:a + int(rng.integers(1, 200))] or b"AC"
                s = s[:a] + unit * int(rng.integers(2, max(3, 8000 // len(unit)))) + s[a:]
            elif kind == 3 and out:
                s = out[int(rng.integers(0, len(out)))][0]            # exact copy of an earlier document
            else:
                s = s[:a] + s[a + int(rng.integers(0, len(s) // 2)):]
        out.append([s if s else b"A"])
    return out


eng = mumemto_amd.Engine(0)
done = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    for case in range(cases):
        if ADV:
            docs = adversarial(rng)
        elif BIG:
            nd_ = int(rng.integers(2, 20))
            docs = synth.pangenome(nd_, int(rng.integers(100000, 1500000) // nd_), float(rng.choice([0.001, 0.005, 0.02])),
                                   seed=int(rng.integers(0, 1 << 30)), indel_rate=float(rng.choice([0, 0.001])))
        elif case % 10 == 9:          # a mid-size pangenome now and then
            docs = synth.pangenome(int(rng.integers(2, 9)), int(rng.integers(3000, 40000)), float(rng.choice([0.002, 0.01, 0.05])),
                                   seed=int(rng.integers(0, 1 << 30)))
        else:
            docs = random_collection(rng)
        p = random_params(rng, len(docs))
        revcomp = bool(rng.integers(0, 2))
        merge = p["max_doc_freq"] == 1 and p["num_distinct"] == len(docs) and bool(rng.integers(0, 2))
        want = O.run(docs, revcomp=revcomp, merge=merge, **p)
        for producer in ("direct", "pfp", "guided"):
            wp = (int(rng.integers(2, 12)), int(rng.choice([3, 5, 7, 11, 13, 16, 20, 37, 100])))
            eng.set_producer(producer, *wp)
            eng.set_docs(docs)
            eng.run(min_match_len=p["min_len"], num_distinct=p["num_distinct"], max_doc_freq=p["max_doc_freq"],
                    max_total_freq=p["max_total_freq"], use_revcomp=revcomp, merge_metadata=merge)
            got = eng.output_text()
            got_thresh = eng.thresholds() if merge else None
            if SHARD:                 # the same run as G shards of the suffix-array positions, concatenated
                G = int(rng.integers(2, 6))
                pieces = []
                for k in range(G):
                    eng.set_scan_shard(k, G)
                    eng.run(min_match_len=p["min_len"], num_distinct=p["num_distinct"], max_doc_freq=p["max_doc_freq"],
                            max_total_freq=p["max_total_freq"], use_revcomp=revcomp, merge_metadata=False)
                    pieces.append(eng.output_text())
                eng.set_scan_shard(0, 1)
                if b"".join(pieces) != got:
                    print("SHARD MISMATCH", seed, case, producer, wp, p, revcomp, G, flush=True)
                    sys.exit(1)
            if got != want.text():
                print("MISMATCH", seed, case, producer, wp, p, revcomp, merge, [[r[:60] for r in d] for d in docs][:3], flush=True)
                sys.exit(1)
            if merge and not np.array_equal(got_thresh, want.thresh()):
                print("THRESH MISMATCH", seed, case, producer, wp, p, flush=True)
                sys.exit(1)
        done += 1
print("fuzz ok: %d collections x 3 producers, seeds %d..%d" % (done, first, first + count - 1))
