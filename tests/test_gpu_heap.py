"""GPU: the device heap behind every engine buffer (mumemto_amd/csrc/pool.hpp)."""
import os

import pytest

import pyoracle as O
from mumemto_amd import synth

pytestmark = pytest.mark.gpu


def test_device_heap_accounting_and_trim():
    """pool.hpp: every engine buffer comes from one growing heap; closing the engines leaves nothing live, trimming gives
    the physical memory back, and a following run maps it again and still matches the oracle."""
    import gc
    import mumemto_amd
    gc.collect()
    L = mumemto_amd.load_library()
    docs = synth.pangenome(6, 30000, 0.01, seed=71)
    want = O.run(docs).text()
    probe = mumemto_amd.Engine(0)               # (device_memory needs a handle; an idle engine holds no buffers)
    base_live = probe.device_memory()["live"]   # engines other tests of this process may still hold
    a, b = mumemto_amd.Engine(0), mumemto_amd.Engine(0)
    for eng in (a, b, a):
        eng.set_docs(docs)
        eng.run()
        assert eng.output_text() == want
    m = a.device_memory()
    # (the peak is the high-water mark of the whole process: tests before this one may have held -- and, since mmt_pool_trim lets go
    # of the library's shared engine too, given back -- far more than is mapped now)
    assert m["mapped"] >= m["live"] > base_live and m["peak"] >= m["live"]
    os.environ["MUMEMTO_LEAN"] = "1"            # stage scratch is released and re-used inside the run
    try:
        c = mumemto_amd.Engine(0)
        c.set_docs(docs)
        c.run(num_distinct=5, max_doc_freq=3)
        assert c.output_text() == O.run(docs, num_distinct=5, max_doc_freq=3).text()
        c.close()
    finally:
        del os.environ["MUMEMTO_LEAN"]
    mapped_before = a.device_memory()["mapped"]
    b.close()
    assert a.device_memory()["live"] > base_live
    a.close()
    assert probe.device_memory()["live"] == base_live and probe.device_memory()["mapped"] == mapped_before
    if base_live == 0:
        L.mmt_pool_trim()
        assert probe.device_memory()["mapped"] == 0
    probe.set_docs(docs)
    probe.run()
    assert probe.output_text() == want and probe.device_memory()["mapped"] > 0
    probe.close()
