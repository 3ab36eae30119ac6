"""Deterministic synthetic pangenomes: the real E. coli /
yeast / HPRC data sets are not available offline, so every benchmark and
large parity test uses `pangenome(N, L, d, seed)`:

  ancestor = L i.i.d. uniform ACGT bases (numpy PCG64, seeded);
  haplotype h = ancestor with every base substituted (uniformly by one of the
  three other bases) independently with probability d.

Optional realism knobs for parity tests: indels, an inversion (exercises the
'-' strand), a tandem duplication (huge LCPs), N runs, lowercase.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def pangenome(n_haps, length, divergence, seed, indel_rate=0.0, inversion=None, tandem=None,
              n_run=None, lowercase_frac=0.0):
    """Returns a list of docs; each doc is a list with one record (bytes)."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=length, dtype=np.uint8)
    docs = []
    for h in range(n_haps):
        hrng = np.random.default_rng([seed, h + 1])
        seq = anc.copy()
        if divergence > 0:
            mut = hrng.random(length) < divergence
            seq[mut] = (seq[mut] + hrng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        arr = _ACGT[seq]
        if indel_rate > 0:
            keep = hrng.random(len(arr)) >= indel_rate / 2
            arr = arr[keep]
            ins = np.nonzero(hrng.random(len(arr)) < indel_rate / 2)[0]
            arr = np.insert(arr, ins, _ACGT[hrng.integers(0, 4, size=len(ins))])
        if inversion and h == inversion[0]:
            a, b = inversion[1], inversion[2]
            comp = {65: 84, 67: 71, 71: 67, 84: 65}
            seg = np.array([comp[c] for c in arr[a:b][::-1]], dtype=np.uint8)
            arr = np.concatenate([arr[:a], seg, arr[b:]])
        if tandem and h == tandem[0]:
            a, b, k = tandem[1], tandem[2], tandem[3]
            arr = np.concatenate([arr[:b]] + [arr[a:b]] * k + [arr[b:]])
        if n_run and h == n_run[0]:
            arr = arr.copy()
            arr[n_run[1]:n_run[2]] = ord("N")
        if lowercase_frac > 0:
            arr = arr.copy()
            low = hrng.random(len(arr)) < lowercase_frac
            arr[low] |= 0x20
        docs.append([arr.tobytes()])
    return docs


def pangenome_subset(n_haps, length, divergence, seed, which):
    """Same haplotypes as pangenome(...)[i] for i in `which`, generated alone
    (per-haplotype RNG streams are independent): SNP-only model."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=length, dtype=np.uint8)
    docs = []
    for h in which:
        hrng = np.random.default_rng([seed, h + 1])
        seq = anc.copy()
        if divergence > 0:
            mut = hrng.random(length) < divergence
            seq[mut] = (seq[mut] + hrng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        docs.append([_ACGT[seq].tobytes()])
    return docs


def write_fasta(path, records, names=None, width=80):
    with open(path, "wb") as f:
        for i, rec in enumerate(records):
            name = names[i] if names else "seq%d" % (i + 1)
            f.write(b">" + name.encode() + b"\n")
            for k in range(0, len(rec), width):
                f.write(rec[k:k + width] + b"\n")


_ASCII_OF_CODE = bytes([65, 67, 71, 84]) + bytes(252)


def ancestor_codes(seed, length):
    """= np.random.default_rng(seed).integers(0, 4, size=length, dtype=np.uint8), several times faster at gigabases: numpy draws a
    bounded uint8 from one byte of the generator's 32-bit output at a time, low byte first, and a range of 4 keeps its top two bits
    (Lemire's method without a rejection: 256 is a multiple of 4) -- so the codes are the top two bits of every byte of the raw
    64-bit stream in little-endian order.  (tests/test_oracle.py holds the two against each other.)"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 0xFFFFFFFFFFFFFFFF, size=(length + 7) // 8, dtype=np.uint64, endpoint=True)
    if raw.dtype.byteorder == ">":
        raw = raw.byteswap()
    return raw.view(np.uint8)[:length] >> 6


def uniform_codes(rng, length):
    """rng.integers(0, 4, size=length, dtype=np.uint8) -- same values AND the same generator state afterwards (numpy takes
    ceil(length / 4) 32-bit outputs for them; an odd count leaves the upper half of a 64-bit output buffered in the generator,
    which a single full-range uint32 draw reproduces)."""
    n32 = (length + 3) // 4
    raw = rng.integers(0, 0xFFFFFFFFFFFFFFFF, size=n32 // 2, dtype=np.uint64, endpoint=True)
    if raw.dtype.byteorder == ">":
        raw = raw.byteswap()
    out = np.empty(n32 * 4, np.uint8)
    out[:(n32 // 2) * 8] = raw.view(np.uint8)
    if n32 & 1:
        last = rng.integers(0, 0xFFFFFFFF, size=1, dtype=np.uint32, endpoint=True)
        out[(n32 // 2) * 8:] = last.astype("<u4").view(np.uint8)
    np.right_shift(out, 6, out=out)
    return out[:length]


def ascii_of_codes(codes):
    """_ACGT[codes] through bytes.translate (a table lookup at copy speed; read-only result)"""
    return np.frombuffer(codes.tobytes().translate(_ASCII_OF_CODE), np.uint8)


def _substitutions(anc, length, divergence, seed, h):
    """(positions, new ASCII bases) of haplotype h, in the order they are applied (a later write to a position wins)"""
    hrng = np.random.default_rng([seed, h + 1])
    k = int(hrng.binomial(length, divergence)) if divergence > 0 else 0
    if not k:
        return np.zeros(0, np.int64), np.zeros(0, np.uint8)
    pos = hrng.integers(0, length, size=k)
    return pos, _ACGT[(anc[pos] + hrng.integers(1, 4, size=k, dtype=np.uint8)) & 3]


def haplotypes_sparse(n_haps, length, divergence, seed, which=None):
    """The same model as `pangenome` (ancestor = L i.i.d. uniform bases, every haplotype = ancestor with substitutions
    at rate d) for collections of gigabases: the substitutions of a haplotype are drawn as k ~ Binomial(L, d) events at
    uniform positions, each replacing the ancestral base by one of the three others -- k instead of L random numbers
    per haplotype, 94 x 64 Mbp in seconds instead of minutes.  Yields (index, uint8 array of ASCII bases) one
    haplotype at a time so that a caller can write or upload each and drop it."""
    anc = ancestor_codes(seed, length)
    anc_ascii = ascii_of_codes(anc)
    for h in (range(n_haps) if which is None else which):
        seq = anc_ascii.copy()
        pos, val = _substitutions(anc, length, divergence, seed, h)
        if len(pos):
            seq[pos] = val
        yield h, seq


def copy_threaded(dst, src, threads=8, piece=1 << 28):
    """dst[:] = src in pieces on a few threads (numpy releases the GIL for the copies): 3 GB in ~0.1 s instead of ~1 s"""
    n = len(src)
    if n <= piece or threads <= 1:
        np.copyto(dst, src)
        return
    from concurrent.futures import ThreadPoolExecutor
    cuts = list(range(0, n, piece)) + [n]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda ab: np.copyto(dst[ab[0]:ab[1]], src[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))


def collection_sparse(n_haps, length, divergence, seed, which=None, threads=8):
    """The haplotypes `haplotypes_sparse` yields for `which`, as ONE flat uint8 array + their lengths (what
    Engine.run_partitioned(flat=...) takes) -- filled by a few threads, a haplotype each: a rank's share of whole genomes
    (13 x 3.05 Gbp) in seconds."""
    which = list(range(n_haps) if which is None else which)
    anc = ancestor_codes(seed, length)
    anc_ascii = ascii_of_codes(anc)
    bases = np.empty(len(which) * length, np.uint8)

    def one(k):
        dst = bases[k * length:(k + 1) * length]
        np.copyto(dst, anc_ascii)
        pos, val = _substitutions(anc, length, divergence, seed, which[k])
        if len(pos):
            dst[pos] = val
    if threads > 1 and len(which) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(threads, len(which))) as ex:
            list(ex.map(one, range(len(which))))
    else:
        for k in range(len(which)):
            one(k)
    return bases, np.full(len(which), length, np.uint64)


def write_fasta_fast(path, bases, name="seq1", width=80):
    """One-record FASTA from a uint8 array, `width` columns, written with two large writes."""
    n = len(bases)
    full = n // width
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        if full:
            body = np.empty((full, width + 1), dtype=np.uint8)
            body[:, :width] = bases[: full * width].reshape(full, width)
            body[:, width] = 10
            f.write(body.tobytes())
        if n % width:
            f.write(bases[full * width:].tobytes() + b"\n")


_COMP_TAB = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP_TAB[_a] = _b


def realistic_ancestor(length, seed):
    """An ancestor with the structures real assemblies carry and i.i.d. sequence lacks (realism knobs):
      * two alpha-satellite-like arrays: a 171-base monomer repeated over 2.3 % and 3.9 % of the length (1.5 and 2.5 Mbp
        at 64 Mbp), every copy 1.5 % diverged from the monomer;
      * twenty microsatellites of period 2 - 6, 10 - 100 kbp each at 64 Mbp (scaled with the length, at least 40 periods),
        0.2 % of their bases substituted;
      * three assembly gaps (runs of N) of 50 kbp, 200 kbp and 1 Mbp at 64 Mbp (scaled, at least 200 bases).
    Returns (uint8 ASCII array, list of (kind, start, end))."""
    rng = np.random.default_rng([seed, 0xA11CE])
    anc = ascii_of_codes(uniform_codes(rng, length)).copy()
    f = length / 64e6
    feats = []
    taken = []

    def place(n):
        n = int(min(n, length // 12))
        for _ in range(200):
            a = int(rng.integers(0, max(1, length - n)))
            if all(a + n <= s or a >= e for s, e in taken):
                taken.append((a, a + n))
                return a, n
        return None, 0

    mono = _ACGT[rng.integers(0, 4, size=171, dtype=np.uint8)]
    for frac in (0.023, 0.039):
        a, n = place(max(171 * 12, int(frac * length)))
        if a is None:
            continue
        arr = np.tile(mono, n // 171 + 1)[:n].copy()
        mut = np.nonzero(rng.random(n) < 0.015)[0]
        arr[mut] = _ACGT[rng.integers(0, 4, size=len(mut))]
        anc[a:a + n] = arr
        feats.append(("satellite", a, a + n))
    for i in range(20):
        period = 2 + i % 5
        n = max(40 * period, int(rng.integers(10_000, 100_001) * f))
        a, n = place(n)
        if a is None:
            continue
        unit = _ACGT[rng.integers(0, 4, size=period, dtype=np.uint8)]
        if len(set(unit.tolist())) == 1:
            unit[0] = _ACGT[(int(np.searchsorted(_ACGT, unit[0])) + 1) & 3]
        arr = np.tile(unit, n // period + 1)[:n].copy()
        mut = np.nonzero(rng.random(n) < 0.002)[0]
        arr[mut] = _ACGT[rng.integers(0, 4, size=len(mut))]
        anc[a:a + n] = arr
        feats.append(("microsatellite%d" % period, a, a + n))
    for gap in (50_000, 200_000, 1_000_000):
        a, n = place(max(200, int(gap * f)))
        if a is None:
            continue
        anc[a:a + n] = ord("N")
        feats.append(("gap", a, a + n))
    return anc, feats


def haplotypes_realistic(n_haps, length, divergence, seed, which=None, indel_rate=1e-4, inversion_every=7):
    """Haplotypes of `realistic_ancestor`: substitutions at rate d (never inside a gap), indels at `indel_rate` per base
    (half deletions, half insertions of 1 - 50 bases), and in every `inversion_every`-th haplotype one inversion of
    0.15 % of the length.  Yields (index, uint8 ASCII array); the lengths differ between haplotypes."""
    anc, _ = realistic_ancestor(length, seed)
    is_n = anc == ord("N")
    for h in (range(n_haps) if which is None else which):
        hrng = np.random.default_rng([seed, h + 1, 0xBEE])
        seq = anc.copy()
        k = int(hrng.binomial(length, divergence)) if divergence > 0 else 0
        if k:
            pos = hrng.integers(0, length, size=k)
            pos = pos[~is_n[pos]]
            cur = np.searchsorted(_ACGT, seq[pos])
            seq[pos] = _ACGT[(cur + hrng.integers(1, 4, size=len(pos))) & 3]
        if inversion_every and h % inversion_every == inversion_every - 1:
            n = max(50, int(0.0015 * length))
            a = int(hrng.integers(0, length - n))
            seq[a:a + n] = _COMP_TAB[seq[a:a + n][::-1]]
        if indel_rate > 0:
            kd = int(hrng.binomial(length, indel_rate / 2))
            ki = int(hrng.binomial(length, indel_rate / 2))
            dpos = np.unique(hrng.integers(0, length, size=kd))
            dlen = hrng.integers(1, 51, size=len(dpos))
            keep = np.ones(length, bool)
            for p, l in zip(dpos.tolist(), dlen.tolist()):
                keep[p:p + l] = False
            seq = seq[keep]
            ipos = np.sort(hrng.integers(0, len(seq), size=ki))
            ilen = hrng.integers(1, 51, size=ki)
            pieces, at = [], 0
            for p, l in zip(ipos.tolist(), ilen.tolist()):
                pieces.append(seq[at:p])
                pieces.append(_ACGT[hrng.integers(0, 4, size=l)])
                at = p
            pieces.append(seq[at:])
            seq = np.concatenate(pieces)
        yield h, seq


def _realistic_to_file(args):
    n_haps, length, divergence, seed, h, path, kw = args
    for _, s in haplotypes_realistic(n_haps, length, divergence, seed, which=[h], **kw):
        np.save(path, s)
        return len(s)


def collection_realistic(n_haps, length, divergence, seed, which=None, procs=1, **kw):
    """The haplotypes `haplotypes_realistic` yields for `which` as ONE flat uint8 array + their (unequal) lengths.  procs > 1:
    a haplotype per worker process (each makes the ancestor again: ~7 bytes of host memory per base and worker at its peak),
    handed over through files in /dev/shm."""
    import os
    which = list(range(n_haps) if which is None else which)
    shm = "/dev/shm"
    if procs <= 1 or len(which) < 2 or not os.path.isdir(shm):
        seqs = [s for _, s in haplotypes_realistic(n_haps, length, divergence, seed, which=which, **kw)]
        return np.concatenate(seqs), np.array([len(s) for s in seqs], np.uint64)
    import multiprocessing as mp
    tag = "mmt_real_%d_%d_" % (os.getpid(), seed)
    paths = [os.path.join(shm, tag + "%03d.npy" % h) for h in which]
    try:
        with mp.get_context("fork").Pool(min(procs, len(which))) as pool:
            lens = pool.map(_realistic_to_file, [(n_haps, length, divergence, seed, h, p, kw) for h, p in zip(which, paths)], chunksize=1)
        bases = np.empty(int(sum(lens)), np.uint8)
        at = 0
        for p, l in zip(paths, lens):
            bases[at:at + l] = np.load(p, mmap_mode="r")
            at += l
            os.unlink(p)
    finally:
        for p in paths:
            if os.path.exists(p):
                os.unlink(p)
    return bases, np.array(lens, np.uint64)
