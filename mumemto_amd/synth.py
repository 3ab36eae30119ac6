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


def haplotypes_sparse(n_haps, length, divergence, seed, which=None):
    """The same model as `pangenome` (ancestor = L i.i.d. uniform bases, every haplotype = ancestor with substitutions
    at rate d) for collections of gigabases: the substitutions of a haplotype are drawn as k ~ Binomial(L, d) events at
    uniform positions, each replacing the ancestral base by one of the three others -- k instead of L random numbers
    per haplotype, 94 x 64 Mbp in seconds instead of minutes.  Yields (index, uint8 array of ASCII bases) one
    haplotype at a time so that a caller can write or upload each and drop it."""
    rng = np.random.default_rng(seed)
    anc = rng.integers(0, 4, size=length, dtype=np.uint8)
    anc_ascii = _ACGT[anc]
    for h in (range(n_haps) if which is None else which):
        hrng = np.random.default_rng([seed, h + 1])
        seq = anc_ascii.copy()
        k = int(hrng.binomial(length, divergence)) if divergence > 0 else 0
        if k:
            pos = hrng.integers(0, length, size=k)
            seq[pos] = _ACGT[(anc[pos] + hrng.integers(1, 4, size=k, dtype=np.uint8)) & 3]
        yield h, seq


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
