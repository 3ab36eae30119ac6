"""CPU: the C-ABI library loads and exports every symbol include/*.h declares
(no compute calls without a GPU) and fails loudly, never silently falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"(?:MUMEMTO_EXPORT|MMT_API)[^;(]*?\b(\w+)\s*\(", src)
    return sorted(set(n for n in names if n != "__attribute__"))


def test_exports_every_declared_symbol():
    import mumemto_amd
    from mumemto_amd import build
    build.build()
    L = mumemto_amd.load_library()
    c_abi = declared_symbols("mumemto.h")
    gpu_abi = declared_symbols("mumemto_gpu.h")
    assert len(c_abi) == 15, c_abi      # the reference's 15 symbols (mumemto.h:56-94)
    for s in c_abi + gpu_abi:
        assert hasattr(L, s), s


def test_no_cpu_fallback_without_gpu(gpu_available):
    import mumemto_amd
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(mumemto_amd.MumemtoError, match="no usable HIP device|no CPU fallback"):
        mumemto_amd.mumemto_mum([[b"ACGT"], [b"ACGT"]])
    with pytest.raises(mumemto_amd.MumemtoError):
        mumemto_amd.Engine(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mumemto_amd")):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in src and "mumemto_oracle" not in src, os.path.join(dirpath, f)
