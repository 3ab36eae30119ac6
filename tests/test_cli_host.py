"""CPU tests of the CLI's host logic (no GPU): option normalisation
(include/pfp_mum.hpp:80-198), input checks (src/ref_builder.cpp:52-138), FASTA
parsing (kseq semantics) and the .lengths writer (src/ref_builder.cpp:193-209).
MUMEMTO_DRY_RUN=1 makes mumemto_exec stop before touching the device."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import build, synth

EXE = os.path.join(os.path.dirname(build.LIB), "..", "bin", "mumemto_exec")


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build()


def run(args, cwd):
    env = dict(os.environ, MUMEMTO_DRY_RUN="1")
    return subprocess.run([EXE] + args, cwd=cwd, env=env, capture_output=True, text=True)


def fields(stdout):
    return dict(kv.split("=") for kv in stdout.split())


def make_inputs(tmp_path, n=5):
    docs = synth.pangenome(n, 500, 0.02, seed=1)
    paths = []
    for i, d in enumerate(docs):
        p = tmp_path / ("g%d.fa" % i)
        synth.write_fasta(str(p), d, width=60)
        paths.append(str(p))
    return docs, paths


def test_parameter_normalisation_matches_reference_rules(tmp_path):
    docs, paths = make_inputs(tmp_path)
    for k, f, F in [(0, 1, 0), (-1, 3, 0), (1, 1, 0), (9, 1, 0), (-9, 1, 0), (2, 0, 100), (0, 2, 100), (0, 2, -1), (0, 0, 1)]:
        r = run(["-o", str(tmp_path / "o"), "-k", str(k), "-f", str(f), "-F", str(F)] + paths, tmp_path)
        assert r.returncode == 0, r.stderr
        nd, mf, mt = O.cli_params(len(paths), k, f, F)
        got = fields(r.stdout)
        assert (int(got["num_distinct"]), int(got["max_doc_freq"]), int(got["max_total_freq"])) == (nd, mf, mt)


def test_flags_and_errors(tmp_path):
    docs, paths = make_inputs(tmp_path)
    got = fields(run(["-o", str(tmp_path / "o"), "-r", "-n", "-b", "-l", "33"] + paths, tmp_path).stdout)
    assert got["revcomp"] == "0" and got["merge"] == "1" and got["anchor"] == "1" and got["binary"] == "1"
    assert got["min_len"] == "33"
    # binary is dropped for multi-MEMs, merging refuses partial / MEM modes
    assert fields(run(["-o", str(tmp_path / "o"), "-b", "-f", "2"] + paths, tmp_path).stdout)["binary"] == "0"
    assert run(["-o", str(tmp_path / "o"), "-M", "-k", "-1"] + paths, tmp_path).returncode == 1
    assert run(["-o", str(tmp_path / "o"), "-M", "-f", "2"] + paths, tmp_path).returncode == 1
    assert run(["-o", str(tmp_path / "o"), paths[0]], tmp_path).returncode == 1          # one input only
    assert run(["-o", str(tmp_path / "o"), paths[0], paths[0]], tmp_path).returncode == 1  # duplicates collapse
    bad = tmp_path / "x.txt"
    bad.write_text(">a\nACGT\n")
    assert run(["-o", str(tmp_path / "o"), paths[0], str(bad)], tmp_path).returncode == 1   # not a FASTA suffix
    empty = tmp_path / "e.fa"
    empty.write_text("")
    r = run(["-o", str(tmp_path / "o"), paths[0], str(empty)], tmp_path)
    assert r.returncode == 1 and "Empty input file" in r.stderr
    # ... the same from the streamed route (the files measured first), and its options that need the collection resident
    senv = dict(os.environ, MUMEMTO_DRY_RUN="1", MUMEMTO_STREAM_INPUT="1")
    r = subprocess.run([EXE, "-o", str(tmp_path / "o"), paths[0], str(empty)], cwd=tmp_path, env=senv, capture_output=True, text=True)
    assert r.returncode == 1 and "Empty input file" in r.stderr
    r = subprocess.run([EXE, "-o", str(tmp_path / "o"), "-K"] + paths, cwd=tmp_path, env=senv, capture_output=True, text=True)
    assert r.returncode == 1 and "streamed" in r.stderr
    assert run(["-o", str(tmp_path / "o"), "-f", "-2"] + paths, tmp_path).returncode == 1


def test_fasta_parsing_and_lengths_file(tmp_path):
    # multi-record, lowercase, CRLF, blank lines, gzip and a FASTQ record
    a = tmp_path / "a.fa"
    a.write_bytes(b">chr1 some comment\nACGTacgt\r\nNNAC\n\n>chr2\nGG\nTTA\n")
    b = tmp_path / "b.fasta.gz"
    with gzip.open(b, "wb") as f:
        f.write(b">only\nACGTTGCA\nACG\n")
    c = tmp_path / "c.fna"
    c.write_bytes(b"@read1 x\nACGTAC\n+\nIIIIII\n@read2\nTTGA\n+read2\nIIII\n")
    r = run(["-o", str(tmp_path / "out")] + [str(a), str(b), str(c)], tmp_path)
    assert r.returncode == 0, r.stderr
    got = fields(r.stdout)
    want = b"ACGTacgtNNAC" + b"GGTTA" + b"ACGTTGCAACG" + b"ACGTAC" + b"TTGA"
    h = 1469598103934665603
    for ch in want:
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(got["bases"]) == len(want) and got["fnv1a"] == "%016x" % h
    lengths = (tmp_path / "out.lengths").read_text().splitlines()
    ra, rb, rc = (os.path.realpath(str(p)) for p in (a, b, c))
    assert lengths == [ra + " * 17", ra + " chr1 12", ra + " chr2 5", rb + " * 11", rb + " only 11",
                       rc + " * 10", rc + " read1 6", rc + " read2 4"]
    # the same files through the streamed route (measured once, read again document by document)
    r2 = subprocess.run([EXE, "-o", str(tmp_path / "out2"), str(a), str(b), str(c)], cwd=tmp_path, capture_output=True, text=True,
                        env=dict(os.environ, MUMEMTO_DRY_RUN="1", MUMEMTO_STREAM_INPUT="1"))
    assert r2.returncode == 0, r2.stderr
    assert fields(r2.stdout) == got and (tmp_path / "out2.lengths").read_text().splitlines() == lengths


def test_filelist_input(tmp_path):
    docs, paths = make_inputs(tmp_path, 3)
    fl = tmp_path / "list.txt"
    fl.write_text("\n".join(p + " 1" for p in paths) + "\n\n")
    got = fields(run(["-i", str(fl), "-o", str(tmp_path / "o")], tmp_path).stdout)
    assert got["docs"] == "3"


def _fnv1a(chunks):
    h = 1469598103934665603
    for values in chunks:
        for v in values:
            h = ((h ^ int(v)) * 1099511628211) % (1 << 64)
    return h


def _fixture_docs(case):
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "newscan", case)
    docs, cur = [], []
    for line in open(os.path.join(g, "input.txt"), "rb").read().split(b"\n"):
        if line.startswith(b"F $"):
            if cur:
                docs.append(cur)
                cur = []
        elif line.startswith(b"F "):
            cur.append(line[2:])
    w, p = open(os.path.join(g, "params.txt")).read().split()
    return g, docs, int(w), int(p)


@pytest.mark.parametrize("case", ["tiny", "three_docs_w4_p11", "three_docs_w10_p100"])
def test_from_parse_checkpoint_rebuilds_the_text(tmp_path, case):
    """-p PREFIX: PREFIX.dict / PREFIX.parse written by the REAL reference parser (golden fixture) expand to exactly
    the text the FASTA path builds (src/pfp_mum.cpp:122-124, include/pfp.hpp:105-129)."""
    import shutil
    g, docs, w, p = _fixture_docs(case)
    shutil.copy(os.path.join(g, "out.dict"), tmp_path / "ck.dict")
    shutil.copy(os.path.join(g, "out.parse"), tmp_path / "ck.parse")
    with open(tmp_path / "ck.lengths", "w") as f:
        for i, d in enumerate(docs):
            f.write("/data/d%d.fa * %d\n" % (i, sum(len(r) for r in d)))
            for j, r in enumerate(d):
                f.write("/data/d%d.fa rec%d %d\n" % (i, j, len(r)))
    r = run(["-p", str(tmp_path / "ck"), "-w", str(w), "-o", str(tmp_path / "out")], tmp_path)
    assert r.returncode == 0, r.stderr
    f = fields(r.stdout)
    text, _ = O.build_text(docs, True)
    assert f["checkpoint"] == "parse" and int(f["text_chars"]) == len(text) == int(f["entries"])
    assert int(f["docs"]) == len(docs) and int(f["num_distinct"]) == len(docs)
    assert int(f["fnv1a"], 16) == _fnv1a([text])
    # a wrong window, a wrong strand setting and a missing lengths file are refused
    assert run(["-p", str(tmp_path / "ck"), "-w", str(w + 1), "-o", str(tmp_path / "out")], tmp_path).returncode == 1
    assert run(["-p", str(tmp_path / "ck"), "-w", str(w), "-r", "-o", str(tmp_path / "out")], tmp_path).returncode == 1
    os.remove(tmp_path / "ck.lengths")
    assert run(["-p", str(tmp_path / "ck"), "-w", str(w), "-o", str(tmp_path / "out")], tmp_path).returncode == 1


def test_arrays_in_checkpoint_reads_the_streams_first_entries(tmp_path):
    """-a PREFIX: 40-bit SA / LCP + BWT files; like the reference (include/read_arrays.hpp:86-104) only the first |T|
    of the |T|+1 entries are used, i.e. every real suffix but the last."""
    import numpy as np
    docs = synth.pangenome(4, 300, 0.02, seed=3)
    text, _ = O.build_text(docs, True)
    sa, lcp, bwt = O.build_stream(text)

    def put40(a):
        a = np.asarray(a, np.uint64)
        return np.stack([(a >> np.uint64(8 * k)) & np.uint64(255) for k in range(5)], axis=1).astype(np.uint8).tobytes()
    (tmp_path / "arr.sa").write_bytes(put40(sa))
    (tmp_path / "arr.lcp").write_bytes(put40(lcp))
    (tmp_path / "arr.bwt").write_bytes(np.asarray(bwt, np.uint8).tobytes())
    with open(tmp_path / "arr.lengths", "w") as f:
        for i, d in enumerate(docs):
            f.write("/data/d%d.fa * %d\n" % (i, len(d[0])))
    r = run(["-a", str(tmp_path / "arr"), "-o", str(tmp_path / "out")], tmp_path)
    assert r.returncode == 0, r.stderr
    f = fields(r.stdout)
    n = len(text)
    assert f["checkpoint"] == "arrays" and int(f["text_chars"]) == n and int(f["entries"]) == n - 1
    assert int(f["fnv1a"], 16) == _fnv1a([sa[1:n], lcp[1:n], bwt[1:n]])
    (tmp_path / "arr.bwt").write_bytes(np.asarray(bwt, np.uint8).tobytes()[: n - 5])      # truncated file
    assert run(["-a", str(tmp_path / "arr"), "-o", str(tmp_path / "out")], tmp_path).returncode == 1
    assert run(["-a", str(tmp_path / "arr"), "-p", str(tmp_path / "arr"), "-o", str(tmp_path / "o")], tmp_path).returncode == 1


def test_in_place_reader_of_plain_files_equals_the_stream_reader(tmp_path):
    """Plain FASTA files are read in cache-sized blocks whose lines go straight to the file's slot (fasta.cpp); everything
    else -- and every file when MUMEMTO_STREAM_READER is set -- goes through the stream reader.  Same bases, same .lengths."""
    rng = np.random.default_rng(5)
    files = []
    for i in range(6):
        recs = []
        for r in range(int(rng.integers(1, 4))):
            seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), size=int(rng.integers(1, 3000))))
            width = int(rng.choice([1, 7, 60, 80, 5000]))
            eol = b"\r\n" if i % 3 == 1 else b"\n"
            lines = eol.join(seq[k:k + width] for k in range(0, len(seq), width))
            recs.append(b">r%d some words%s%s%s" % (r, eol, lines, b"" if (i == 5 and r == 0) else eol + (b"\n" if r % 2 else b"")))
        p = tmp_path / ("f%d.fa" % i)
        p.write_bytes((b"junk before the first header\n" if i == 2 else b"") + b"".join(recs))
        files.append(str(p))
    outs = []
    # ... and MUMEMTO_STREAM_INPUT=1: the files measured once, then every document read again when it is asked for, in order
    # (cli_main.cpp::StreamedInput: what mumemto_exec does with a collection that does not fit the host)
    for env in ({}, {"MUMEMTO_STREAM_READER": "1"}, {"MUMEMTO_STREAM_INPUT": "1"}):
        r = subprocess.run([EXE, "-o", str(tmp_path / ("o%d" % len(outs)))] + files, cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, MUMEMTO_DRY_RUN="1", **env))
        assert r.returncode == 0, r.stderr
        outs.append((fields(r.stdout), (tmp_path / ("o%d.lengths" % len(outs))).read_text()))
    for k in (1, 2):
        assert outs[0][0]["bases"] == outs[k][0]["bases"] and outs[0][0]["fnv1a"] == outs[k][0]["fnv1a"]
        assert outs[0][1] == outs[k][1] and int(outs[0][0]["bases"]) > 0


def test_chunked_reader_gives_the_same_bases_as_the_arena(tmp_path):
    """The command line's one-shot runs send the bases to the device through fixed-size chunks instead of holding the collection
    in host memory (fasta.hpp, chunked mode of ReadHooks).  MUMEMTO_DRY_RUN_CHUNK runs that reader into host vectors: with
    chunks of 1 .. 64 bytes every line end, '\\r' and record boundary falls on a chunk boundary somewhere -- the hash over all
    documents' bases and the document lengths must be the arena reader's, for plain FASTA, CRLF line ends, several records per
    file, empty lines, and a FASTQ file (which falls back to the stream reader in the middle of a chunked document)."""
    import random
    rng = random.Random(5)

    def seq(n):
        return "".join(rng.choice("ACGTN") for _ in range(n))
    a = tmp_path / "a.fa"
    a.write_text(">r1 first\n" + "\n".join(seq(60) for _ in range(9)) + "\n" + seq(17) + "\n>r2\n\n" + seq(60) + "\n\n" + seq(3) + "\n")
    b = tmp_path / "b.fa"
    b.write_bytes((">crlf\r\n" + "\r\n".join(seq(rng.choice([1, 2, 59, 60, 61])) for _ in range(25)) + "\r\n>x\r\n" + seq(5) + "\r").encode())
    c = tmp_path / "c.fa"
    c.write_text(">one_line\n" + seq(1000))
    d = tmp_path / "d.fa"        # (FASTQ content under a FASTA name: the suffix is what the command line checks)
    d.write_text("".join("@q%d\n%s\n+\n%s\n" % (i, seq(40), "I" * 40) for i in range(6)))
    e = tmp_path / "e.fa"      # many line widths around the copier's 32 bytes, some CRLF, some empty lines, several records
    lines = []
    for i in range(4000):
        if i % 700 == 0:
            lines.append(">rec%d some description" % i)
        wd = rng.choice([0, 1, 2, 30, 31, 32, 33, 34, 62, 63, 64, 65, 66, 95, 96, 97, 200])
        lines.append(seq(wd) + ("\r" if rng.random() < 0.2 else ""))
    e.write_bytes("\n".join(lines).encode())
    paths = [str(a), str(b), str(c), str(d), str(e)]
    want = fields(run(["-o", str(tmp_path / "o")] + paths, tmp_path).stdout)
    assert int(want["bases"]) > 2000

    # the truth, from a reader of a few lines (kseq's rules: include/kseq.h:176-216): the arena reader -- whose sequence lines go
    # through the 32-byte copier (fasta.cpp copy_sequence_lines_avx2) --, the same without it, and the chunked one must all hash to it
    def bases_of(path):
        out, in_seq, quality_left = bytearray(), False, 0
        for line in open(path, "rb").read().split(b"\n"):
            if line.endswith(b"\r"):
                line = line[:-1]
            if quality_left > 0:
                quality_left -= len(line)
                continue
            if line[:1] in (b">", b"@"):
                in_seq, rec = True, len(out)
            elif line[:1] == b"+" and in_seq:
                in_seq, quality_left = False, len(out) - rec
            elif in_seq:
                out += line
        return bytes(out)
    h = 1469598103934665603
    total = 0
    for q in paths:
        for byte in bases_of(q):
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        total += len(bases_of(q))
    assert (int(want["bases"]), want["fnv1a"]) == (total, "%016x" % h)
    r = subprocess.run([EXE, "-o", str(tmp_path / "o")] + paths, cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, MUMEMTO_DRY_RUN="1", MUMEMTO_NO_AVX2="1"))
    assert fields(r.stdout)["fnv1a"] == want["fnv1a"]
    for chunk in (1, 2, 3, 7, 59, 60, 61, 64, 4096):
        r = subprocess.run([EXE, "-o", str(tmp_path / "o")] + paths, cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, MUMEMTO_DRY_RUN="1", MUMEMTO_DRY_RUN_CHUNK=str(chunk)))
        assert r.returncode == 0, r.stderr
        got = fields(r.stdout)
        assert (got["docs"], got["bases"], got["fnv1a"]) == (want["docs"], want["bases"], want["fnv1a"]), (chunk, got, want)
    # the three plain files alone (no fallback in between)
    want = fields(run(["-o", str(tmp_path / "o")] + paths[:3], tmp_path).stdout)
    r = subprocess.run([EXE, "-o", str(tmp_path / "o")] + paths[:3], cwd=tmp_path, capture_output=True, text=True,
                       env=dict(os.environ, MUMEMTO_DRY_RUN="1", MUMEMTO_DRY_RUN_CHUNK="5"))
    assert fields(r.stdout)["fnv1a"] == want["fnv1a"]
