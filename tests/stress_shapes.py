"""GPU box helper: adversarial shapes (time and parity) -- homopolymers, exact copies, long tandem repeats, two-letter
texts, one huge document next to tiny ones.  usage: stress_shapes.py [scale]   (each case is checked against the oracle)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE]
import numpy as np
import pyoracle as O
import mumemto_amd
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
# The CPU oracle (like the reference's mem_finder it restates) is quadratic on homopolymers: 6 s at 100 k characters,
# 54 s at 300 k.  Above 300 k the two producers are compared with each other instead.
USE_ORACLE = scale <= 300_000
rng = np.random.default_rng(5)
def rnd(n, alphabet=b"ACGT"):
    return np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)].tobytes()
base = rnd(scale)
cases = {
    "homopolymer docs": [[b"A" * scale], [b"A" * (scale // 2) + b"C" + b"A" * (scale // 2)], [b"A" * (scale - 7)]],
    "exact copies x6": [[base]] * 6,
    "tandem repeat (period 37)": [[rnd(37) * (scale // 37)], [rnd(500) + rnd(37) * (scale // 74)], [base[: scale // 2]]],
    "two-letter text": [[rnd(scale, b"AT")], [rnd(scale, b"AT")], [rnd(scale // 3, b"AT")]],
    "one big, many tiny": [[base]] + [[base[i * 50: i * 50 + 40]] for i in range(40)],
    "N runs": [[base[:1000] + b"N" * (scale // 2) + base[1000:2000]], [base[:1500] + b"N" * (scale // 3)], [b"N" * 5000 + base[:900]]],
    "period-2 and period-3": [[b"AC" * (scale // 2)], [b"ACG" * (scale // 3)], [b"AC" * (scale // 4) + b"ACG" * (scale // 6)]],
}
eng = mumemto_amd.Engine(0)
for name, docs in cases.items():
    for kw in (dict(num_distinct=0, max_doc_freq=1), dict(num_distinct=2, max_doc_freq=3, max_total_freq=40)):
        first = None
        for producer in ("pfp", "direct"):
            eng.set_producer(producer)
            eng.set_docs(docs)
            t = time.perf_counter()
            eng.run(min_match_len=20, **kw)
            dt = time.perf_counter() - t
            got = eng.output_text()
            if USE_ORACLE:
                ok = got == O.run(docs, min_len=20, **kw).text()
            else:
                ok = first is None or got == first
                first = got
            print("%-28s %-7s %-42s %8.1f ms  rows %-7d %s" % (name, producer, kw, dt * 1e3, eng.L.mmt_num_rows(eng.h),
                                                             "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                sys.exit(1)
print("stress ok")
