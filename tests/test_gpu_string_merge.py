"""String-based merge end to end on the GPU: `mumemto_exec -M` per partition, extract_mums,
`mumemto_exec` on the `#`-terminated MUM strings, re-threshold -- every file byte-compared with what the reference's
own tools wrote for the same documents (tests/golden/string_merge, tests/golden/make_string_merge.py)."""
import os
import shutil
import subprocess

import pytest

from mumemto_amd import build, merge_mums, synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(build.LIB), "..", "bin")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "string_merge")
CASES = sorted(os.listdir(GOLD))


def read(path):
    with open(path, "rb") as f:
        return f.read()


def lengths_table(path):
    # the CLI writes canonical paths (src/ref_builder.cpp:193-209), the fixture relative ones
    return [[os.path.basename(t[0])] + t[1:] for t in (ln.split() for ln in read(path).decode().splitlines())]


def cli(args):
    r = subprocess.run([os.path.join(BIN, "mumemto_exec")] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.fixture
def case_dir(tmp_path, request):
    dst = tmp_path / request.param
    shutil.copytree(os.path.join(GOLD, request.param), dst)
    cwd = os.getcwd()
    os.chdir(dst)
    yield [[int(x) for x in ln.split(",")] for ln in read("groups.txt").decode().split()]
    os.chdir(cwd)


@pytest.mark.parametrize("case_dir", CASES, indirect=True)
def test_partitions_extract_and_merge_equal_the_reference_files(case_dir):
    groups = case_dir
    files = []
    for g, members in enumerate(groups):
        cli(["doc%d.fa" % i for i in members] + ["-M", "-o", "q%d" % g])
        merge_mums.sort_partition("q%d" % g)                      # into the order q<g>.thresh is written in
        for ext in (".mums", ".thresh", ".thresh_rev"):
            assert read("q%d%s" % (g, ext)) == read("p%d%s" % (g, ext)), (g, ext)
        assert lengths_table("q%d.lengths" % g) == lengths_table("p%d.lengths" % g)
        files.append("q%d.mums" % g)
    merge_mums.main(merge_mums.parse_arguments(files + ["-o", "ours"]))     # extract, MUMs of MUMs on the GPU, fold
    for ext in (".mums", ".thresh", ".thresh_rev"):
        assert read("ours" + ext) == read("merged" + ext), ext
    assert lengths_table("ours.lengths") == lengths_table("merged.lengths")
    assert not [f for f in os.listdir(".") if "_temp_merged" in f or f.startswith("q") and f.endswith("_mums.fa")]


def test_merged_partitions_of_a_larger_collection_equal_the_direct_run(tmp_path):
    docs = synth.pangenome(9, 200000, 0.004, seed=77, indel_rate=0.001, inversion=(4, 50000, 90000))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for i, d in enumerate(docs):
            synth.write_fasta("h%d.fa" % i, d)
        groups = [[0, 1, 2], [4, 3, 5], [6, 7, 8]]                # partition 1 starts with the inverted haplotype
        for g, members in enumerate(groups):
            cli(["h%d.fa" % i for i in members] + ["-M", "-o", "part%d" % g])
            merge_mums.sort_partition("part%d" % g)
        n = merge_mums.main(merge_mums.parse_arguments(["part%d.mums" % g for g in range(3)] + ["-o", "all.mums"]))
        cli(["h%d.fa" % i for g in groups for i in g] + ["-o", "direct"])
        merge_mums.sort_partition("direct")
        merged, direct = read("all.mums").splitlines(), read("direct.mums").splitlines()
        assert n == len(merged) > 1000
        assert set(merged) <= set(direct) and len(direct) - len(merged) <= 1      # see test_string_merge_host.py
        assert merged == [ln for ln in direct if ln in set(merged)]
        # merging the merged output with a fourth partition works on the files this tool wrote (dynamic updating,
        # README.md:141 of the reference)
        assert os.path.getsize("all.thresh") == os.path.getsize("all.thresh_rev") == 2 * sum(
            int(ln.split(b"\t")[0]) + 1 for ln in merged)
    finally:
        os.chdir(cwd)
