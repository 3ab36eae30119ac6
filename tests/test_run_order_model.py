"""CPU: the rule the suffix sorter uses for suffixes that begin with a long run of one symbol (sorter.cpp refine_runs,
kernels.hip k_run_keys), checked as a statement about strings -- no device involved.  A suffix c^j X (X0 != c) is ordered among
the suffixes that begin with at least h copies of c by: class (X0 < c before X0 > c), then j ascending in the first class and
descending in the second, then X.  The model sorts every such suffix of random texts by that key and by plain comparison."""
import random


def test_order_inside_a_bucket_of_long_runs():
    rng = random.Random(11)
    h = 4
    for _ in range(300):
        # symbols 1..5, 0 is the end of the text (smaller than every symbol, as the padding behind the dictionary is)
        n = rng.randint(30, 120)
        t = []
        while len(t) < n:
            c = rng.randint(1, 5)
            t += [c] * rng.choice([1, 1, 2, 3, h, h + 1, 2 * h, 3 * h + 1])
        t = t[:n] + [0]
        for c in range(1, 6):
            bucket = [p for p in range(n) if t[p:p + h] == [c] * h]
            if len(bucket) < 2:
                continue

            def key(p):
                j = 0
                while t[p + j] == c:
                    j += 1
                x = t[p + j:]
                cls = 1 if x[0] > c else 0
                return (cls, j if cls == 0 else -j, x)
            assert sorted(bucket, key=key) == sorted(bucket, key=lambda p: t[p:]), (t, c)
