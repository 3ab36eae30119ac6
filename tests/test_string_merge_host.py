"""String-based merge, host side: `extract_mums` and the re-threshold step of
mumemto_amd/merge_mums.py against the files the REAL reference tools wrote (tests/golden/string_merge, made by
tests/golden/make_string_merge.py from src/extract_mums.cpp and mumemto/merge_mums.py).  No GPU: the MUMs-of-MUMs
table is part of the fixture and handed over with -m."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mumemto_amd import merge_mums, mumsio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "string_merge")
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mumemto_amd", "bin")
CASES = sorted(os.listdir(GOLD))


def n_parts(case):
    return len([f for f in os.listdir(os.path.join(GOLD, case)) if f.startswith("p") and f.endswith(".mums")])


@pytest.fixture
def workdir(tmp_path, request):
    case = request.param
    dst = tmp_path / case
    shutil.copytree(os.path.join(GOLD, case), dst)
    cwd = os.getcwd()
    os.chdir(dst)                     # the .lengths files name the FASTA files relative to the case directory
    yield case
    os.chdir(cwd)


def read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("workdir", CASES, indirect=True)
def test_extract_mums_writes_what_the_reference_tool_writes(workdir):
    exe = os.path.join(BIN, "extract_mums")
    assert os.path.exists(exe), "run python -m mumemto_amd.build"
    for g in range(n_parts(workdir)):
        subprocess.check_call([exe, "-m", "p%d.mums" % g, "-o", "ours_%d" % g])
        assert read("ours_%d.fa" % g) == read("p%d_mums.fa" % g)
        subprocess.check_call([exe, "p%d" % g, "-t", "-o", "bare_%d.fa" % g])        # positional prefix, no terminator
        assert read("bare_%d.fa" % g) == read("p%d_mums.fa" % g).replace(b"#", b"")


def test_extract_mums_refuses_partial_rows_and_missing_files(tmp_path):
    exe = os.path.join(BIN, "extract_mums")
    (tmp_path / "a.fa").write_text(">r\nACGTACGTACGTACGTACGTACGTACGT\n")
    (tmp_path / "x.lengths").write_text("%s 28\n%s 28\n" % (tmp_path / "a.fa", tmp_path / "a.fa"))
    (tmp_path / "x.mums").write_text("20\t,3\t+,+\n")
    assert subprocess.run([exe, "-m", str(tmp_path / "x.mums")], capture_output=True).returncode == 1
    assert subprocess.run([exe, "-m", str(tmp_path / "nope.mums")], capture_output=True).returncode == 1
    (tmp_path / "y.mums").write_text("20\t3,3\t+,+\n")
    assert subprocess.run([exe, "-m", str(tmp_path / "y.mums")], capture_output=True).returncode == 1   # no y.lengths


@pytest.mark.parametrize("workdir", CASES, indirect=True)
def test_merge_step_reproduces_the_reference_outputs(workdir):
    files = ["p%d.mums" % g for g in range(n_parts(workdir))]
    merge_mums.main(merge_mums.parse_arguments(["-m", "mom.mums", "-o", "ours"] + files))
    for ext in (".mums", ".thresh", ".thresh_rev", ".lengths"):
        assert read("ours" + ext) == read("merged" + ext), ext
    merge_mums.main(merge_mums.parse_arguments(["-m", "mom", "-o", "ours_bin.bumbl"] + files))
    assert read("ours_bin.bumbl") == read("merged_bin.bumbl")
    for a, b in zip(mumsio.read_bumbl("ours_bin.bumbl"), mumsio.read_mums("merged.mums")):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("workdir", CASES, indirect=True)
def test_merged_rows_are_the_direct_run_of_the_union(workdir):
    """README.md:132 of the reference: identical to a run on the union -- up to the row that touches the end of the
    first document, which extract_mums truncates (src/extract_mums.cpp:106)."""
    merged = read("merged.mums").splitlines()
    direct = read("direct.mums").splitlines()
    assert set(merged) <= set(direct)
    assert len(direct) - len(merged) <= 1
    assert merged == [ln for ln in direct if ln in set(merged)]          # same (first-offset) order


def test_pieces_of_a_match_that_spans_terminators():
    # collection 0 holds three MUM records of lengths 24, 3 and 30 (+1 for '#'): '#' at 24, 28, 59
    term = np.array([24, 28, 59])
    lengths = np.array([60], np.uint32)
    starts = np.array([[2, 100]], np.int64)
    strands = np.array([[True, False]])
    seg_len, seg_start, seg_fwd = merge_mums.split_at_terminators((lengths, starts, strands), term)
    # pieces: [2,24) = 22 kept, [25,28) = 3 dropped, [29,59) = 30 kept, [60,62) = 2 dropped
    assert seg_len.tolist() == [22, 30]
    assert seg_start[:, 0].tolist() == [2, 29]
    assert seg_start[:, 1].tolist() == [100 + 60 - 22, 100 + 60 - 57]   # reverse strand: measured from the far end
    assert seg_fwd.tolist() == [[True, False], [True, False]]
    # a match without a terminator inside is passed through whatever its length
    seg_len, seg_start, _ = merge_mums.split_at_terminators((np.array([5], np.uint32), np.array([[30, 7]], np.int64),
                                                             np.array([[True, True]])), term)
    assert seg_len.tolist() == [5] and seg_start.tolist() == [[30, 7]]


def test_lengths_files_are_joined_like_the_reference_does(tmp_path):
    (tmp_path / "a.lengths").write_text("x/a.fa 10\nx/b.fa 12\n")
    (tmp_path / "b.lengths").write_text("x/c.fa * 7\nx/c.fa r0 3\nx/c.fa r1 4\n")
    (tmp_path / "a.mums").write_text("")
    (tmp_path / "b.mums").write_text("")
    args = merge_mums.parse_arguments([str(tmp_path / "a.mums"), str(tmp_path / "b.mums"), "-o", str(tmp_path / "o")])
    merge_mums.merge_lengths(args)
    assert (tmp_path / "o.lengths").read_text() == ("x/a.fa * 10\nx/a.fa a.fa 10\nx/b.fa * 12\nx/b.fa b.fa 12\n"
                                                    "x/c.fa * 7\nx/c.fa r0 3\nx/c.fa r1 4")
    assert [r.tolist() for r in merge_mums.record_lengths(str(tmp_path / "o.lengths"))] == [[10], [12], [3, 4]]
