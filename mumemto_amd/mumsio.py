"""PREFIX.mums / PREFIX.bumbl tables as NumPy arrays (host-side I/O for the merge tools).

Formats as the reference reads and writes them (mumemto/utils.py:69-87,627-632,655-665; include/mumsio.hpp:105-194):
  .mums   one row per line: `length <TAB> off_0,off_1,... <TAB> s_0,s_1,...` (empty offset = absent, strands + / -)
  .bumbl  u16 flags (bit 13 partial, bit 14 collinear blocks, bit 15 32-bit lengths) | u64 n_docs | u64 n_rows |
          lengths (u16 or u32) | offsets i64 [n_rows][n_docs] | strand bits, row-major, most significant bit first
"""
import numpy as np

FLAG_PARTIAL = 1 << 13
FLAG_BLOCKS = 1 << 14
FLAG_LENGTH32 = 1 << 15


def read_mums(path):
    """-> (lengths u32 [n], starts i64 [n, N] with -1 = absent, strands bool [n, N])"""
    lengths, starts, strands = [], [], []
    with open(path, "rb") as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            lengths.append(int(parts[0]))
            starts.append([int(x) if x else -1 for x in parts[1].split(b",")])
            strands.append([x == b"+" for x in parts[2].split(b",")])
    n_docs = len(starts[0]) if starts else 0
    return (np.array(lengths, np.uint32), np.array(starts, np.int64).reshape(len(lengths), n_docs),
            np.array(strands, bool).reshape(len(lengths), n_docs))


def read_bumbl(path):
    raw = np.fromfile(path, np.uint8)
    flags = int(raw[:2].view(np.uint16)[0])
    n_docs, n_rows = (int(x) for x in raw[2:18].view(np.uint64))
    pos = 18
    if flags & FLAG_LENGTH32:
        lengths = raw[pos:pos + 4 * n_rows].view(np.uint32).copy()
        pos += 4 * n_rows
    else:
        lengths = raw[pos:pos + 2 * n_rows].view(np.uint16).astype(np.uint32)
        pos += 2 * n_rows
    cells = n_rows * n_docs
    starts = raw[pos:pos + 8 * cells].view(np.int64).reshape(n_rows, n_docs).copy()
    pos += 8 * cells
    bits = np.unpackbits(raw[pos:pos + (cells + 7) // 8])[:cells]
    return lengths, starts, bits.astype(bool).reshape(n_rows, n_docs)


def read_rows(path):
    return read_bumbl(path) if path.endswith(".bumbl") else read_mums(path)


def write_mums(path, lengths, starts, strands):
    with open(path, "w") as f:
        for length, row, srow in zip(lengths.tolist(), starts.tolist(), strands.tolist()):
            f.write("%d\t%s\t%s\n" % (length, ",".join(map(str, row)), ",".join("+" if s else "-" for s in srow)))


def write_bumbl(path, lengths, starts, strands):
    lengths = np.ascontiguousarray(lengths)
    starts = np.ascontiguousarray(starts, np.int64)
    flags = (FLAG_PARTIAL if (starts == -1).any() else 0) | (FLAG_LENGTH32 if lengths.dtype == np.uint32 else 0)
    with open(path, "wb") as f:
        f.write(np.uint16(flags).tobytes())
        f.write(np.uint64(starts.shape[1] if starts.ndim == 2 else 0).tobytes())
        f.write(np.uint64(len(lengths)).tobytes())
        f.write(lengths.tobytes())
        f.write(starts.tobytes())
        f.write(np.packbits(np.ascontiguousarray(strands, bool)).tobytes())
