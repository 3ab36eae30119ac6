"""ctypes front-end of oracle/libmumemto_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from mumemto_amd/.  See oracle/mumemto_oracle.h for the reference
file:line each entry point restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmumemto_oracle.so")


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is mounted)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mumemto_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libmumemto_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class ScanParams(C.Structure):
    _fields_ = [
        ("min_len", C.c_int64),
        ("num_distinct", C.c_int64),
        ("max_doc_freq", C.c_int64),
        ("max_total_freq", C.c_int64),
        ("revcomp", C.c_int32),
        ("merge", C.c_int32),
    ]


class Partition(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_docs", C.c_int64),
        ("length", C.c_void_p),
        ("offsets", C.c_void_p),
        ("strands", C.c_void_p),
        ("nb", C.c_void_p),
        ("nb_len", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.mmo_text_length.restype = C.c_int64
        L.mmo_text_length.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.mmo_build_text.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.mmo_build_stream.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mmo_doc_array.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.mmo_scan.restype = C.c_void_p
        L.mmo_scan.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_int64, C.POINTER(ScanParams)]
        L.mmo_result_free.argtypes = [C.c_void_p]
        for name in ("mmo_num_intervals", "mmo_num_accepted", "mmo_num_rows", "mmo_num_occ", "mmo_thresh_len"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.mmo_get_intervals.argtypes = [C.c_void_p, C.c_void_p]
        L.mmo_get_accepted.argtypes = [C.c_void_p, C.c_void_p]
        L.mmo_get_mum_rows.argtypes = [C.c_void_p] * 4
        L.mmo_get_mem_rows.argtypes = [C.c_void_p] * 6
        L.mmo_format_text.restype = C.c_void_p
        L.mmo_format_text.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.mmo_format_bumbl.restype = C.c_void_p
        L.mmo_format_bumbl.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.mmo_get_thresh.argtypes = [C.c_void_p, C.c_void_p]
        L.mmo_format_thresh.restype = C.c_void_p
        L.mmo_format_thresh.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.mmo_free.argtypes = [C.c_void_p]
        L.mmo_anchor_merge.restype = C.c_void_p
        L.mmo_anchor_merge.argtypes = [C.POINTER(Partition), C.c_int64]
        L.mmo_merged_rows.restype = C.c_int64
        L.mmo_merged_rows.argtypes = [C.c_void_p]
        L.mmo_merged_docs.restype = C.c_int64
        L.mmo_merged_docs.argtypes = [C.c_void_p]
        L.mmo_merged_get.argtypes = [C.c_void_p] * 5
        L.mmo_merged_free.argtypes = [C.c_void_p]
        L.mmo_build_stream_pfp.restype = C.c_int
        L.mmo_build_stream_pfp.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        L.mmo_run_job_pfp.restype = C.c_int64
        L.mmo_run_job_pfp.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ScanParams), C.c_int64, C.c_int64,
                                      C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.mmo_run_job.restype = C.c_int64
        L.mmo_run_job.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ScanParams),
                                  C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def docs_to_bases(docs):
    """docs: list of docs, each a list of record byte strings -> (bases u8, doc_len i64)."""
    lens = np.array([sum(len(r) for r in d) for d in docs], dtype=np.int64)
    flat = b"".join(b"".join(d) for d in docs)
    bases = np.frombuffer(flat, dtype=np.uint8).copy() if flat else np.zeros(1, np.uint8)
    return bases, lens


def build_text(docs, revcomp=True):
    L = lib()
    bases, lens = docs_to_bases(docs)
    n = L.mmo_text_length(_p(lens), len(docs), int(revcomp))
    text = np.zeros(max(n, 1), np.uint8)
    doc_start = np.zeros(len(docs) + 1, np.int64)
    L.mmo_build_text(_p(bases), _p(lens), len(docs), int(revcomp), _p(text), _p(doc_start))
    return text[:n], doc_start


def build_stream(text):
    """Returns (sa, lcp, bwt) of the n+1-entry stream (entry 0 = end sentinel)."""
    L = lib()
    n = len(text)
    sa = np.zeros(n + 1, np.int64)
    lcp = np.zeros(n + 1, np.int64)
    bwt = np.zeros(n + 1, np.uint8)
    t = np.ascontiguousarray(text if n else np.zeros(1, np.uint8))
    rc = L.mmo_build_stream(_p(t), n, _p(sa), _p(lcp), _p(bwt))
    if rc:
        raise RuntimeError("mmo_build_stream failed rc=%d" % rc)
    return sa, lcp, bwt


class ScanResult:
    def __init__(self, handle, n_docs, mummode):
        self.h = handle
        self.n_docs = n_docs
        self.mummode = mummode

    def __del__(self):
        if self.h:
            lib().mmo_result_free(self.h)
            self.h = None

    def intervals(self):
        L = lib()
        out = np.zeros((L.mmo_num_intervals(self.h), 4), np.int64)
        if len(out):
            L.mmo_get_intervals(self.h, _p(out))
        return out

    def accepted(self):
        L = lib()
        out = np.zeros((L.mmo_num_accepted(self.h), 4), np.int64)
        if len(out):
            L.mmo_get_accepted(self.h, _p(out))
        return out

    def mum_rows(self):
        L = lib()
        n = L.mmo_num_rows(self.h)
        length = np.zeros(max(n, 1), np.uint32)
        off = np.zeros((max(n, 1), self.n_docs), np.int64)
        st = np.zeros((max(n, 1), self.n_docs), np.uint8)
        L.mmo_get_mum_rows(self.h, _p(length), _p(off), _p(st))
        return length[:n], off[:n], st[:n]

    def mem_rows(self):
        L = lib()
        n = L.mmo_num_rows(self.h)
        t = L.mmo_num_occ(self.h)
        length = np.zeros(max(n, 1), np.uint32)
        occ = np.zeros(n + 1, np.int64)
        off = np.zeros(max(t, 1), np.int64)
        docs = np.zeros(max(t, 1), np.int64)
        st = np.zeros(max(t, 1), np.uint8)
        L.mmo_get_mem_rows(self.h, _p(length), _p(occ), _p(off), _p(docs), _p(st))
        return length[:n], occ, off[:t], docs[:t], st[:t]

    def _take(self, ptr, nbytes):
        step = 1 << 30                      # ctypes.string_at takes a C int
        base = C.cast(ptr, C.c_void_p).value
        data = b"".join(C.string_at(base + o, min(step, nbytes - o)) for o in range(0, nbytes, step)) if nbytes else b""
        lib().mmo_free(ptr)
        return data

    def text(self):
        n = C.c_int64()
        ptr = lib().mmo_format_text(self.h, C.byref(n))
        return self._take(ptr, n.value)

    def bumbl(self):
        n = C.c_int64()
        ptr = lib().mmo_format_bumbl(self.h, C.byref(n))
        return self._take(ptr, n.value)

    def thresh(self):
        L = lib()
        out = np.zeros(max(L.mmo_thresh_len(self.h), 1), np.uint16)
        n = L.mmo_thresh_len(self.h)
        if n:
            L.mmo_get_thresh(self.h, _p(out))
        return out[:n]

    def thresh_file(self, rev=False):
        n = C.c_int64()
        ptr = lib().mmo_format_thresh(self.h, int(rev), C.byref(n))
        return np.frombuffer(self._take(ptr, n.value * 2), np.uint16).copy()


def scan(sa, lcp, bwt, doc_start, min_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0,
         revcomp=True, merge=False):
    L = lib()
    n_docs = len(doc_start) - 1
    m = len(sa)
    doc = np.zeros(m, np.int32)
    ds = np.ascontiguousarray(doc_start, dtype=np.int64)
    L.mmo_doc_array(_p(sa), m, _p(ds), n_docs, _p(doc))
    p = ScanParams(min_len, num_distinct if num_distinct else n_docs, max_doc_freq, max_total_freq,
                   int(revcomp), int(merge))
    h = L.mmo_scan(_p(sa), _p(lcp), _p(bwt), _p(doc), m, _p(ds), n_docs, C.byref(p))
    return ScanResult(h, n_docs, max_doc_freq == 1)


def run(docs, **kw):
    """docs -> ScanResult, the whole path on the CPU."""
    revcomp = kw.get("revcomp", True)
    text, doc_start = build_text(docs, revcomp)
    sa, lcp, bwt = build_stream(text)
    return scan(sa, lcp, bwt, doc_start, **kw)


def cli_params(n_docs, k=0, f=1, F=0):
    """BuildOptions::set_parameters (include/pfp_mum.hpp:149-198): CLI flag
    normalisation -> (num_distinct, max_doc_freq, max_total_freq)."""
    nd = k
    if nd < -n_docs:
        nd = 2
    elif nd <= 0:
        nd = n_docs + nd
    elif nd == 1:
        nd = 2
    elif nd >= n_docs:
        nd = n_docs
    mf = F
    if mf < -n_docs or mf == 1:
        mf = 0
    elif mf < 0:
        mf = n_docs + mf
    if f > 0 and (mf == 0 or mf > f * n_docs):
        mf = f * n_docs
    return nd, f, mf


def anchor_merge(parts):
    """parts: list of (length u32[n], offsets i64[n,nd], strands u8[n,nd], nb u16[L0+1])."""
    L = lib()
    arr = (Partition * len(parts))()
    keep = []
    for i, (length, off, st, nb) in enumerate(parts):
        length = np.ascontiguousarray(length, np.uint32)
        off = np.ascontiguousarray(off, np.int64).reshape(len(length), -1)
        st = np.ascontiguousarray(st, np.uint8).reshape(len(length), -1)
        nb = np.ascontiguousarray(nb, np.uint16)
        keep += [length, off, st, nb]
        arr[i] = Partition(len(length), off.shape[1], _p(length).value, _p(off).value, _p(st).value,
                           _p(nb).value, len(nb))
    h = L.mmo_anchor_merge(arr, len(parts))
    n, nd = L.mmo_merged_rows(h), L.mmo_merged_docs(h)
    length = np.zeros(max(n, 1), np.uint32)
    off = np.zeros((max(n, 1), nd), np.int64)
    st = np.zeros((max(n, 1), nd), np.uint8)
    nb = np.zeros(len(parts[0][3]), np.uint16)
    L.mmo_merged_get(h, _p(length), _p(off), _p(st), _p(nb))
    L.mmo_merged_free(h)
    return length[:n], off[:n], st[:n], nb


def build_stream_pfp(text, w=10, p=100):
    """The stream by way of the prefix-free parse (the reference's default route): (sa, lcp, bwt, stats) with
    stats = (phrases, distinct phrases, dictionary bytes, stream entries).  Must equal build_stream(text)."""
    L = lib()
    text = np.ascontiguousarray(text, dtype=np.uint8)
    n = len(text)
    sa = np.zeros(n + 1, np.int64); lcp = np.zeros(n + 1, np.int64); bwt = np.zeros(n + 1, np.uint8)
    st = np.zeros(4, np.int64)
    rc = L.mmo_build_stream_pfp(_p(text), n, w, p, _p(sa), _p(lcp), _p(bwt), _p(st))
    if rc:
        raise RuntimeError("mmo_build_stream_pfp failed (%d)" % rc)
    return sa, lcp, bwt, tuple(int(x) for x in st)


def run_job_timed(docs, min_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, revcomp=True, pfp=None):
    """Whole job on one core; returns (text_len, stage seconds[3], .mums bytes).  pfp = (w, p): the stream is produced
    through the prefix-free parse (the reference's default route) instead of a suffix sort of the whole text (-g)."""
    L = lib()
    bases, lens = docs_to_bases(docs)
    p = ScanParams(min_len, num_distinct if num_distinct else len(docs), max_doc_freq, max_total_freq,
                   int(revcomp), 0)
    sec = (C.c_double * 3)()
    out = C.c_void_p()
    n = C.c_int64()
    if pfp:
        tl = L.mmo_run_job_pfp(_p(bases), _p(lens), len(docs), C.byref(p), int(pfp[0]), int(pfp[1]), sec, C.byref(out),
                               C.byref(n))
    else:
        tl = L.mmo_run_job(_p(bases), _p(lens), len(docs), C.byref(p), sec, C.byref(out), C.byref(n))
    if tl < 0:
        raise RuntimeError("oracle job failed (%d)" % tl)
    data = C.string_at(out, n.value)
    L.mmo_free(out)
    return tl, list(sec), data
