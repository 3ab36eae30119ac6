"""GPU: the drop-in claim, shown with the reference's OWN client code compiled
unmodified (oracle/Makefile, target ref) against this repository's libmumemto.so:
  * oracle/_ref/dropin_wrapper  -- mumemto_library/mumemto.hpp (header-only RAII wrapper over the C ABI,
    with the reference's mumemto.h / mumemto_api.hpp / mumsio.hpp)
  * oracle/_ref/_mumemto_core*.so -- python_bindings/src/mumemto_pybind.cpp (pybind11 over the C++ API)
Their results must equal the oracle's."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import pyoracle as O
from mumemto_amd import synth
from mumsfile import format_mums

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _need(path):
    if not glob.glob(path):
        pytest.skip("oracle/_ref artefact not built (needs the reference tree at build time): " + path)
    return glob.glob(path)[0]


def test_reference_cxx_wrapper_runs_on_this_library(tmp_path):
    exe = _need(os.path.join(REF, "dropin_wrapper"))
    docs = synth.pangenome(4, 6000, 0.01, seed=41, inversion=(1, 1000, 1500))
    stdin = b"\n".join(b",".join(d) for d in docs) + b"\n"
    out = tmp_path / "w.mums"
    r = subprocess.run([exe, "mum", "20", "1", str(out)], input=stdin, capture_output=True)
    assert r.returncode == 0, r.stderr
    want = O.run(docs)
    wl, wo, ws = want.mum_rows()
    assert r.stdout.split() == [b"4", str(len(wl)).encode()]
    assert out.read_bytes() == format_mums(wl, wo, ws)          # strict MUMs: wrapper format == serialize_mum
    out2 = tmp_path / "w.mems"
    r = subprocess.run([exe, "mem", "20", "1", str(out2)], input=stdin, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert int(r.stdout.split()[1]) == len(O.run(docs, max_doc_freq=2).mem_rows()[0])


def test_reference_pybind_module_runs_on_this_library():
    _need(os.path.join(REF, "_mumemto_core*.so"))
    import mumemto_amd  # noqa: F401  (loads torch's HIP runtime first, see binding.py)
    mumemto_amd.load_library()
    sys.path.insert(0, REF)
    try:
        import _mumemto_core as core
    finally:
        sys.path.remove(REF)
    docs = synth.pangenome(3, 5000, 0.02, seed=42)
    seqs = [[r.decode() for r in d] for d in docs]
    res = core.mum(seqs, 20)
    want = O.run(docs)
    wl, wo, ws = want.mum_rows()
    assert res.num_docs() == 3 and res.num_matches() == len(wl) and len(res) == len(wl)
    for i in range(len(wl)):
        length, offsets, strands = res.match_at(i)
        assert length == wl[i] and np.array_equal(offsets, wo[i]) and np.array_equal(strands.astype(np.uint8), ws[i])
    mres = core.mem(seqs, 20, True, 0, 0, 2)
    ml, mocc, mo, md, ms = O.run(docs, max_doc_freq=2).mem_rows()
    assert mres.num_matches() == len(ml)
    for i in range(0, len(ml), max(1, len(ml) // 50)):
        length, offsets, ids, strands = mres.match_at(i)
        a, b = mocc[i], mocc[i + 1]
        assert length == ml[i] and np.array_equal(offsets, mo[a:b]) and np.array_equal(ids, md[a:b])
    with pytest.raises(Exception):
        core.mem(seqs, 20, True, 0, 0, 1)      # f <= 1 -> invalid_argument, as in the reference
