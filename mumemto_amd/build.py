"""In-tree build of libmumemto.so (hipcc, gfx950 only) and the CLI tools.

`python -m mumemto_amd.build` or `mumemto_amd.build.build()`.  Objects go to
mumemto_amd/csrc/_build/, products to mumemto_amd/lib/ and mumemto_amd/bin/
(git-ignored; they travel to the GPU box with the snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(SRC, "_build")
LIB_DIR = os.path.join(HERE, "lib")
BIN_DIR = os.path.join(HERE, "bin")
LIB = os.path.join(LIB_DIR, "libmumemto.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result",
            "--offload-arch=" + ARCH]

LIB_SOURCES = ["kernels.hip", "prims.hip", "parse_lcp.hip", "pfp_kernels.hip", "rows_kernels.hip", "merge_kernels.hip", "engine.cpp", "sorter.cpp", "pfp.cpp", "guided.cpp", "guided_kernels.hip", "pool.cpp", "dist.cpp", "merge.cpp", "partitioned.cpp", "api.cpp",
               "cxx_api.cpp", "fasta.cpp", "options.cpp"]
TOOLS = {"mumemto_exec": ["cli_main.cpp"], "anchor_merge": ["merge_main.cpp"]}
HOST_TOOLS = {"extract_mums": ["extract_mums_main.cpp", "fasta.cpp"]}     # no device code: plain g++


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".hpp", ".h"))]
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc)]
    return hs


def _compile(src):
    path = os.path.join(SRC, src)
    obj = os.path.join(OBJ, src + ".o")
    if not os.path.exists(path):
        return None
    if _newer(obj, [path] + _headers()):
        cmd = [HIPCC] + CXXFLAGS + ["-c", path, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-6000:]))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(verbose=False):
    for d in (OBJ, LIB_DIR, BIN_DIR):
        os.makedirs(d, exist_ok=True)
    tool_sources = [s for v in TOOLS.values() for s in v]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, LIB_SOURCES + tool_sources))
    lib_objs = [o for o in objs[: len(LIB_SOURCES)] if o]
    if _newer(LIB, lib_objs + [os.path.join(SRC, "libmumemto.map")]):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-o", LIB] + lib_objs + [
            "-Wl,-soname,libmumemto.so", "-Wl,--version-script=" + os.path.join(SRC, "libmumemto.map"), "-lz", "-ldl"]
        subprocess.check_call(cmd)
    for tool, srcs in TOOLS.items():
        tobjs = [os.path.join(OBJ, s + ".o") for s in srcs if os.path.exists(os.path.join(OBJ, s + ".o"))]
        if not tobjs:
            continue
        out = os.path.join(BIN_DIR, tool)
        if _newer(out, tobjs + lib_objs):
            # the tools use the engine classes directly (hidden symbols of the .so): link the objects in
            subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-o", out] + tobjs + lib_objs + ["-lz", "-ldl"])
    for tool, srcs in HOST_TOOLS.items():
        paths = [os.path.join(SRC, f) for f in srcs]
        out = os.path.join(BIN_DIR, tool)
        if _newer(out, paths + _headers()):
            subprocess.check_call([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-pthread", "-o", out] + paths + ["-lz"])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(verbose=True)
