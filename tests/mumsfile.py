"""Small readers for the reference's file formats (used by tests only)."""
import numpy as np


def parse_mums(data):
    """.mums text -> (length u32[n], offsets i64[n,N] (-1 absent), strands u8[n,N])."""
    rows = [l for l in data.decode().split("\n") if l]
    if not rows:
        return np.zeros(0, np.uint32), np.zeros((0, 0), np.int64), np.zeros((0, 0), np.uint8)
    L, O_, S = [], [], []
    for l in rows:
        a, b, c = l.split("\t")[:3]
        L.append(int(a))
        O_.append([int(x) if x else -1 for x in b.split(",")])
        S.append([1 if x == "+" else 0 for x in c.split(",")])
    return np.array(L, np.uint32), np.array(O_, np.int64), np.array(S, np.uint8)


def format_mums(length, offsets, strands):
    """merged-output style (mumsio.hpp:281-294 serialize_mum): no blanks."""
    out = []
    for l, o, s in zip(length, offsets, strands):
        out.append("%d\t%s\t%s\n" % (l, ",".join(str(int(x)) for x in o), ",".join("+" if x else "-" for x in s)))
    return "".join(out).encode()
