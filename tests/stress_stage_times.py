"""GPU box helper: per-stage times of the adversarial shapes that stress_shapes.py shows to be slow."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import numpy as np
import mumemto_amd
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
rng = np.random.default_rng(5)
def rnd(n, alphabet=b"ACGT"):
    return np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)].tobytes()
base = rnd(scale)
cases = {
    "homopolymer docs": [[b"A" * scale], [b"A" * (scale // 2) + b"C" + b"A" * (scale // 2)], [b"A" * (scale - 7)]],
    "N runs": [[base[:1000] + b"N" * (scale // 2) + base[1000:2000]], [base[:1500] + b"N" * (scale // 3)], [b"N" * 5000 + base[:900]]],
    "period-2 and period-3": [[b"AC" * (scale // 2)], [b"ACG" * (scale // 3)], [b"AC" * (scale // 4) + b"ACG" * (scale // 6)]],
    "N gaps in a pangenome": None,
}
names = ["text", "suffix sort", "lcp+bwt", "scan", "verify", "rows", "format", "total"]
eng = mumemto_amd.Engine(0)
for name, docs in cases.items():
    if docs is None:      # 8 haplotypes x scale, each with a 5 % run of N at a different place
        from mumemto_amd import synth
        docs = synth.pangenome(8, scale, 0.005, 9)
        docs = [[d[0][: (i + 1) * scale // 10] + b"N" * (scale // 20) + d[0][(i + 1) * scale // 10:]] for i, d in enumerate(docs)]
    for producer in ("pfp", "direct"):
        eng.set_producer(producer)
        eng.set_docs(docs)
        eng.run(min_match_len=20, num_distinct=0, max_doc_freq=1)
        ms = eng.stage_ms()
        print("%-24s %-7s " % (name, producer) + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, ms)), flush=True)
        if producer == "pfp":
            print("    pfp:", eng.pfp_counts(), [round(float(x), 1) for x in eng.pfp_stage_ms()], flush=True)

# satellite array: a 171-bp monomer repeated 10,000 times with 2 % differences between the copies, 0.5 % between the
# haplotypes (centromeric higher-order repeats), embedded in random sequence
def satellite(h):
    r = np.random.default_rng(77)
    mono = r.integers(0, 4, 171, dtype=np.uint8)
    arr = np.tile(mono, 10000)
    mut = r.random(arr.size) < 0.02
    arr[mut] = (arr[mut] + r.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
    hr = np.random.default_rng(1000 + h)
    mut = hr.random(arr.size) < 0.005
    arr[mut] = (arr[mut] + hr.integers(1, 4, int(mut.sum()), dtype=np.uint8)) & 3
    return np.frombuffer(b"ACGT", np.uint8)[arr].tobytes()
docs = [[rnd(200000) + satellite(h) + rnd(200000)] for h in range(8)]
for producer in ("pfp", "direct"):
    eng.set_producer(producer)
    eng.set_docs(docs)
    eng.run(min_match_len=20, num_distinct=0, max_doc_freq=1)
    print("%-24s %-7s " % ("satellite arrays x8", producer) + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, eng.stage_ms())), flush=True)
    if producer == "pfp":
        print("    pfp:", eng.pfp_counts(), [round(float(x), 1) for x in eng.pfp_stage_ms()], flush=True)
    first = eng.output_text() if producer == "pfp" else first
    assert eng.output_text() == first
