"""ctypes binding of libmumemto.so (include/mumemto.h + include/mumemto_gpu.h)."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libmumemto.so")
_lib = None


class MumemtoError(RuntimeError):
    pass


def library_path():
    return _LIB


class DocView(C.Structure):
    _fields_ = [("records", C.POINTER(C.c_char_p)), ("num_records", C.c_size_t)]


class MumView(C.Structure):
    _fields_ = [("length", C.c_uint32), ("offsets", C.POINTER(C.c_int64)), ("strands", C.POINTER(C.c_uint8))]


class MemView(C.Structure):
    _fields_ = [("length", C.c_uint32), ("occurrences", C.c_size_t), ("offsets", C.POINTER(C.c_int64)),
                ("seq_ids", C.POINTER(C.c_size_t)), ("strands", C.POINTER(C.c_uint8))]


class Params(C.Structure):
    """mmt_params (include/mumemto_gpu.h)."""
    _fields_ = [("min_match_len", C.c_uint32), ("num_distinct", C.c_uint64), ("max_doc_freq", C.c_int64),
                ("max_total_freq", C.c_int64), ("use_revcomp", C.c_uint8), ("merge_metadata", C.c_uint8)]


# mmt_doc_supplier (mumemto_gpu.h): int (*)(void* user, uint64_t doc, uint8_t* dst, uint64_t len)
DOC_SUPPLIER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64)


class Partition(C.Structure):  # mmt_partition (mumemto_gpu.h); thresh_bits 0 / 16 / 32
    _fields_ = [("n_rows", C.c_uint64), ("n_docs", C.c_uint64), ("length", C.c_void_p), ("offsets", C.c_void_p),
                ("strands", C.c_void_p), ("thresh", C.c_void_p), ("thresh_len", C.c_uint64),
                ("thresh_on_device", C.c_uint8), ("rows_on_device", C.c_uint8), ("thresh_bits", C.c_uint8)]


class DevicePartition:
    """One partition's MUM rows and anchor thresholds as HBM addresses (what the multi-GPU exchange
    produces): length u32[n_rows], offsets i64[n_rows, n_docs], strands u8[n_rows, n_docs],
    thresh i16/u16[thresh_len].  `keepalive` holds whatever owns the memory."""

    def __init__(self, n_rows, n_docs, length_ptr, offsets_ptr, strands_ptr, thresh_ptr, thresh_len, keepalive=None,
                 thresh_bits=16):
        self.thresh_bits = int(thresh_bits)
        self.n_rows, self.n_docs = int(n_rows), int(n_docs)
        self.length_ptr, self.offsets_ptr, self.strands_ptr = int(length_ptr), int(offsets_ptr), int(strands_ptr)
        self.thresh_ptr, self.thresh_len = int(thresh_ptr), int(thresh_len)
        self.keepalive = keepalive


C_ABI_SYMBOLS = [
    "mumemto_last_error", "mumemto_mum", "mumemto_mem", "num_docs", "doc_record_offsets", "record_lengths",
    "num_mums", "mum_at", "mum_free", "num_docs_mem", "doc_record_offsets_mem", "record_lengths_mem", "num_mems",
    "mem_at", "mem_free",
]
GPU_ABI_SYMBOLS = [
    "mmt_last_error", "mmt_engine_create", "mmt_engine_destroy", "mmt_engine_set_input_device",
    "mmt_engine_set_input_host", "mmt_engine_run", "mmt_num_rows", "mmt_num_docs", "mmt_rows_mum", "mmt_num_occ",
    "mmt_rows_mem", "mmt_output_text", "mmt_output_bumbl", "mmt_thresh_len", "mmt_copy_thresh", "mmt_thresh_device",
    "mmt_text_length", "mmt_copy_text", "mmt_copy_sa", "mmt_copy_lcp", "mmt_copy_bwt", "mmt_num_candidates",
    "mmt_copy_candidates", "mmt_stage_ms", "mmt_column_bytes", "mmt_anchor_merge", "mmt_merged_rows",
    "mmt_merged_docs", "mmt_merged_get", "mmt_merged_sort_like_direct", "mmt_merged_text", "mmt_merged_free",
    "mmt_engine_set_producer", "mmt_abi_version", "mmt_engine_set_row_tap", "mmt_text_sink_digest", "mmt_kmer_in_share", "mmt_row_tap_counts", "mmt_row_tap_get", "mmt_kmer_positions", "mmt_producer_used", "mmt_producer_expanded", "mmt_producer_stats", "mmt_engine_parse_only", "mmt_pfp_counts", "mmt_pfp_run_refined", "mmt_pfp_copy_dict",
    "mmt_pfp_copy_parse", "mmt_pfp_stage_ms", "mmt_engine_run_partitioned", "mmt_partitions_used",
    "mmt_copy_merged_thresh", "mmt_rows_mum_device", "mmt_merged_device", "mmt_engine_set_text_host",
    "mmt_engine_set_stream_host", "mmt_is_wide", "mmt_scan_ranges", "mmt_copy_sa64", "mmt_engine_set_stream_host40",
    "mmt_device_memory", "mmt_engine_run_files", "mmt_anchor_merge_min_len", "mmt_pool_trim", "mmt_pool_set_reserve",
    "mmt_engine_set_scan_shard", "mmt_merged_from_rows", "mmt_anchor_merge_by_ranges", "mmt_dist_merge_ranges", "mmt_fold_slice_bounds", "mmt_comm_unique_id", "mmt_comm_create", "mmt_comm_destroy", "mmt_comm_loopback", "mmt_comm_selftest", "mmt_dist_merge",
    "mmt_dist_gather_text", "mmt_merged_write_text", "mmt_sort_pieces", "mmt_engine_keep_columns", "mmt_columns_kept",
    "mmt_stream_stats", "mmt_engine_release_columns", "mmt_copy_thresh32", "mmt_thresh_device32", "mmt_engine_set_text_sink",
    "mmt_engine_run_supplied",
]


def load_library():
    """Loads libmumemto.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise MumemtoError("libmumemto.so is not built (%s): run `python -m mumemto_amd.build`; "
                           "mumemto_amd has no CPU fallback" % _LIB)
    if not os.environ.get("MUMEMTO_NO_TORCH"):
        # PyTorch-ROCm ships its own libamdhip64 under the same soname.  If torch is (or will be)
        # in this process it must be loaded first, so that both share ONE HIP runtime and device
        # pointers / streams can be exchanged; loading ours first leaves torch without devices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(_LIB)
    L.mumemto_last_error.restype = C.c_char_p
    L.mmt_last_error.restype = C.c_char_p
    L.mumemto_mum.argtypes = [C.POINTER(DocView), C.c_size_t, C.c_uint32, C.c_uint8, C.c_size_t, C.c_uint8,
                              C.POINTER(C.c_void_p)]
    L.mumemto_mem.argtypes = [C.POINTER(DocView), C.c_size_t, C.c_uint32, C.c_uint8, C.c_size_t, C.c_size_t,
                              C.c_size_t, C.c_uint8, C.POINTER(C.c_void_p)]
    for f in ("num_docs", "num_mums", "num_docs_mem", "num_mems"):
        getattr(L, f).restype = C.c_size_t
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("doc_record_offsets", "record_lengths", "doc_record_offsets_mem", "record_lengths_mem"):
        getattr(L, f).restype = C.POINTER(C.c_size_t)
        getattr(L, f).argtypes = [C.c_void_p]
    L.mum_at.restype = MumView
    L.mum_at.argtypes = [C.c_void_p, C.c_size_t]
    L.mem_at.restype = MemView
    L.mem_at.argtypes = [C.c_void_p, C.c_size_t]
    L.mum_free.argtypes = [C.c_void_p]
    L.mem_free.argtypes = [C.c_void_p]

    L.mmt_engine_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.mmt_engine_destroy.argtypes = [C.c_void_p]
    L.mmt_engine_set_input_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mmt_engine_set_input_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mmt_engine_set_text_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]
    L.mmt_engine_set_stream_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_size_t, C.c_int]
    L.mmt_engine_run.argtypes = [C.c_void_p, C.POINTER(Params)]
    for f in ("mmt_num_rows", "mmt_num_docs", "mmt_num_occ", "mmt_thresh_len", "mmt_num_candidates",
              "mmt_merged_rows", "mmt_merged_docs"):
        getattr(L, f).restype = C.c_size_t
        getattr(L, f).argtypes = [C.c_void_p]
    L.mmt_text_length.restype = C.c_uint64
    L.mmt_text_length.argtypes = [C.c_void_p]
    L.mmt_rows_mum.argtypes = [C.c_void_p] * 4
    L.mmt_rows_mem.argtypes = [C.c_void_p] * 6
    L.mmt_output_text.restype = C.c_void_p
    L.mmt_output_text.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.mmt_output_bumbl.restype = C.c_void_p
    L.mmt_output_bumbl.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.mmt_copy_thresh.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_thresh_device.restype = C.c_void_p
    L.mmt_thresh_device.argtypes = [C.c_void_p]
    for f in ("mmt_copy_text", "mmt_copy_sa", "mmt_copy_lcp", "mmt_copy_bwt", "mmt_copy_candidates"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_copy_sa64.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_is_wide.argtypes = [C.c_void_p]
    L.mmt_scan_ranges.restype = C.c_size_t
    L.mmt_scan_ranges.argtypes = [C.c_void_p]
    L.mmt_device_memory.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mmt_engine_set_stream_host40.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                               C.c_void_p, C.c_size_t, C.c_int]
    L.mmt_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.mmt_column_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.mmt_engine_set_producer.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]
    L.mmt_producer_used.argtypes = [C.c_void_p]
    L.mmt_producer_expanded.argtypes = [C.c_void_p]
    L.mmt_producer_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_text_sink_digest.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mmt_kmer_in_share.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mmt_engine_set_row_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.mmt_row_tap_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mmt_row_tap_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mmt_kmer_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.POINTER(C.c_uint64)]
    L.mmt_engine_parse_only.argtypes = [C.c_void_p, C.c_uint8, C.c_uint32, C.c_uint32]
    L.mmt_pfp_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mmt_pool_set_reserve.argtypes = [C.c_ulonglong]
    L.mmt_pool_set_reserve.restype = None
    L.mmt_pfp_run_refined.argtypes = [C.c_void_p]
    L.mmt_pfp_run_refined.restype = C.c_longlong
    L.mmt_pfp_copy_dict.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_pfp_copy_parse.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_pfp_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.mmt_engine_run_supplied.argtypes = [C.c_void_p, DOC_SUPPLIER, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Params)]
    L.mmt_engine_run_partitioned.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Params),
                                             C.c_uint64]
    L.mmt_engine_run_files.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(Params), C.c_char_p,
                                       C.c_uint64, C.POINTER(C.c_double)]
    L.mmt_merged_from_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                       C.c_size_t, C.POINTER(C.c_void_p)]
    L.mmt_comm_unique_id.argtypes = [C.c_void_p]
    L.mmt_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.mmt_comm_destroy.argtypes = [C.c_void_p]
    L.mmt_comm_loopback.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_comm_selftest.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.mmt_dist_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.mmt_dist_gather_text.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.mmt_partitions_used.restype = C.c_size_t
    L.mmt_partitions_used.argtypes = [C.c_void_p]
    L.mmt_copy_merged_thresh.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_anchor_merge.argtypes = [C.c_void_p, C.POINTER(Partition), C.c_size_t, C.POINTER(C.c_void_p)]
    L.mmt_anchor_merge_by_ranges.argtypes = [C.c_void_p, C.POINTER(Partition), C.c_size_t, C.c_int, C.c_uint32,
                                             C.POINTER(C.c_void_p)]
    L.mmt_dist_merge_ranges.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.mmt_fold_slice_bounds.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64)]
    L.mmt_anchor_merge_min_len.argtypes = [C.c_void_p, C.POINTER(Partition), C.c_size_t, C.c_uint32,
                                           C.POINTER(C.c_void_p)]
    L.mmt_merged_get.argtypes = [C.c_void_p] * 5
    L.mmt_merged_device.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
    L.mmt_rows_mum_device.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3
    L.mmt_merged_sort_like_direct.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_merged_write_text.argtypes = [C.c_void_p, C.c_char_p]
    L.mmt_engine_keep_columns.argtypes = [C.c_void_p, C.c_int]
    L.mmt_columns_kept.argtypes = [C.c_void_p]
    L.mmt_stream_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_sort_pieces.restype = C.c_size_t
    L.mmt_sort_pieces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mmt_merged_text.restype = C.c_void_p
    L.mmt_merged_text.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.mmt_merged_free.argtypes = [C.c_void_p]
    L.mmt_engine_release_columns.argtypes = [C.c_void_p, C.c_int]
    L.mmt_engine_set_text_sink.argtypes = [C.c_void_p, C.c_char_p]
    L.mmt_copy_thresh32.argtypes = [C.c_void_p, C.c_void_p]
    L.mmt_thresh_device32.restype = C.c_void_p
    L.mmt_thresh_device32.argtypes = [C.c_void_p]
    _lib = L
    return L


def _bytes_at(ptr, n):
    """n bytes at ptr as `bytes`; ctypes.string_at takes a C int, a MEM-mode output can exceed 2 GiB"""
    if not n:
        return b""
    step = 1 << 30
    if n <= step:
        return C.string_at(ptr, n)
    base = C.cast(ptr, C.c_void_p).value
    return b"".join(C.string_at(base + o, min(step, n - o)) for o in range(0, n, step))


def _check(rc, gpu=True):
    if rc != 0:
        L = load_library()
        msg = (L.mmt_last_error() if gpu else L.mumemto_last_error()) or b""
        raise MumemtoError("libmumemto rc=%d: %s" % (rc, msg.decode(errors="replace")))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _doc_views(sequences):
    views = (DocView * max(len(sequences), 1))()
    keep = []
    for i, doc in enumerate(sequences):
        recs = [r if isinstance(r, bytes) else r.encode() for r in doc]
        arr = (C.c_char_p * max(len(recs), 1))(*recs)
        keep.append(arr)
        views[i].records = arr
        views[i].num_records = len(recs)
    return views, keep


# ---- drop-in API (mirror of mumemto_library/mumemto_api.hpp:29-57) --------------
def mumemto_mum(sequences, min_match_len=20, use_revcomp=True, num_distinct=0, use_gsacak=False):
    """Multi-MUMs of `sequences` (list of docs, each a list of record strings).
    Returns dict(lengths=u32[n], offsets=i64[n,N] (-1 absent), strands=u8[n,N] (1='+'),
    record_lengths=list of per-doc record lengths) -- through the C ABI of mumemto.h."""
    L = load_library()
    views, keep = _doc_views(sequences)
    out = C.c_void_p()
    rc = L.mumemto_mum(views, len(sequences), min_match_len, int(bool(use_revcomp)), num_distinct,
                       int(bool(use_gsacak)), C.byref(out))
    _check(rc, gpu=False)
    try:
        n, N = L.num_mums(out), L.num_docs(out)
        lengths = np.zeros(n, np.uint32)
        offsets = np.zeros((n, N), np.int64)
        strands = np.zeros((n, N), np.uint8)
        for i in range(n):
            v = L.mum_at(out, i)
            lengths[i] = v.length
            offsets[i] = np.ctypeslib.as_array(v.offsets, shape=(N,))
            strands[i] = np.ctypeslib.as_array(v.strands, shape=(N,))
        dro = L.doc_record_offsets(out)
        rl = L.record_lengths(out)
        rec = [[rl[k] for k in range(dro[d], dro[d + 1])] for d in range(N)]
        return dict(lengths=lengths, offsets=offsets, strands=strands, record_lengths=rec)
    finally:
        L.mum_free(out)


def mumemto_mem(sequences, min_match_len=20, use_revcomp=True, num_distinct=0, max_total_freq=0, max_doc_freq=2,
                use_gsacak=False):
    """Multi-MEMs; returns a list of dict(length, offsets, seq_ids, strands) + record_lengths."""
    L = load_library()
    views, keep = _doc_views(sequences)
    out = C.c_void_p()
    rc = L.mumemto_mem(views, len(sequences), min_match_len, int(bool(use_revcomp)), num_distinct, max_total_freq,
                       max_doc_freq, int(bool(use_gsacak)), C.byref(out))
    _check(rc, gpu=False)
    try:
        n, N = L.num_mems(out), L.num_docs_mem(out)
        mems = []
        for i in range(n):
            v = L.mem_at(out, i)
            k = v.occurrences
            mems.append(dict(length=int(v.length),
                             offsets=np.ctypeslib.as_array(v.offsets, shape=(k,)).copy(),
                             seq_ids=np.ctypeslib.as_array(v.seq_ids, shape=(k,)).astype(np.int64),
                             strands=np.ctypeslib.as_array(v.strands, shape=(k,)).copy()))
        dro = L.doc_record_offsets_mem(out)
        rl = L.record_lengths_mem(out)
        rec = [[rl[k] for k in range(dro[d], dro[d + 1])] for d in range(N)]
        return dict(mems=mems, record_lengths=rec)
    finally:
        L.mem_free(out)


# ---- device-resident engine ----------------------------------------------------------
class Engine:
    """One GPU's hot path (include/mumemto_gpu.h)."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.mmt_engine_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.mmt_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_docs(self, docs):
        """docs: list of docs, each a list of record bytes (host memory)."""
        lens = np.array([sum(len(r) for r in d) for d in docs], dtype=np.uint64)
        flat = b"".join(b"".join(d) for d in docs)
        bases = np.frombuffer(flat, dtype=np.uint8) if flat else np.zeros(1, np.uint8)
        _check(self.L.mmt_engine_set_input_host(self.h, _p(bases), _p(lens), len(docs)))

    def set_input_device(self, dev_ptr, doc_len, keepalive=None):
        """dev_ptr: address of the concatenated bases in HBM (e.g. tensor.data_ptr())."""
        lens = np.ascontiguousarray(doc_len, dtype=np.uint64)
        self._keep = keepalive
        _check(self.L.mmt_engine_set_input_device(self.h, C.c_void_p(dev_ptr), _p(lens), len(lens)))

    def set_text(self, text, doc_len, use_revcomp=True):
        """Hand over the text T itself (bytes; UPPER(F) '$' [revcomp '$'] per document): the `-p` checkpoint."""
        t = np.frombuffer(bytes(text), np.uint8) if len(text) else np.zeros(1, np.uint8)
        lens = np.ascontiguousarray(doc_len, np.uint64)
        _check(self.L.mmt_engine_set_text_host(self.h, _p(t), C.c_uint64(len(text)), _p(lens), len(lens), int(use_revcomp)))

    def set_stream(self, sa, lcp, bwt, doc_len, use_revcomp=True):
        """Hand over SA / LCP / BWT of the real suffixes (possibly a prefix of the stream): the `-a` checkpoint."""
        sa = np.ascontiguousarray(sa, np.uint32); lcp = np.ascontiguousarray(lcp, np.uint32)
        bwt = np.ascontiguousarray(bwt, np.uint8); lens = np.ascontiguousarray(doc_len, np.uint64)
        _check(self.L.mmt_engine_set_stream_host(self.h, _p(sa), _p(lcp), _p(bwt), C.c_uint64(len(sa)), _p(lens), len(lens),
                                                 int(use_revcomp)))

    def run(self, min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, use_revcomp=True,
            merge_metadata=False):
        p = Params(min_match_len, num_distinct, max_doc_freq, max_total_freq, int(use_revcomp), int(merge_metadata))
        _check(self.L.mmt_engine_run(self.h, C.byref(p)))

    def run_partitioned(self, docs, max_text_chars=0, min_match_len=20, use_revcomp=True, num_distinct=0,
                        max_doc_freq=1, max_total_freq=0, flat=None, merge_metadata=False):
        """Host-resident docs of any size: one suffix array when the text fits the device (40-bit positions beyond
        2^32 characters), otherwise anchor partitions + merge (strict multi-MUMs only).  `flat` = (uint8 array of
        the concatenated bases, uint64 array of document lengths) skips the Python-side concatenation."""
        if flat is not None:
            bases, lens = flat
            lens = np.ascontiguousarray(lens, dtype=np.uint64)
        else:
            lens = np.array([sum(len(r) for r in d) for d in docs], dtype=np.uint64)
            joined = b"".join(b"".join(d) for d in docs)
            bases = np.frombuffer(joined, dtype=np.uint8) if joined else np.zeros(1, np.uint8)
        p = Params(min_match_len, num_distinct, max_doc_freq, max_total_freq, int(use_revcomp), int(merge_metadata))
        _check(self.L.mmt_engine_run_partitioned(self.h, _p(bases), _p(lens), len(lens), C.byref(p), max_text_chars))
        return int(self.L.mmt_partitions_used(self.h))

    def run_supplied(self, lens, supplier, min_match_len=20, use_revcomp=True, num_distinct=0, max_doc_freq=1,
                     max_total_freq=0, merge_metadata=False):
        """The documents supplied one at a time: `supplier(d, dst)` writes the bases of document d into the uint8 array dst
        (len = lens[d], page-locked memory of the engine).  For collections that do not fit the host as bytes; always one
        text (mmt_engine_run_supplied)."""
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        failure = []

        def _fill(_user, d, dst, n):
            try:
                supplier(int(d), np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_uint8)), shape=(int(n),)))
                return 0
            except BaseException as exc:                       # (an exception must not unwind through the C frames)
                failure.append(exc)
                return 1
        cb = DOC_SUPPLIER(_fill)
        p = Params(min_match_len, num_distinct, max_doc_freq, max_total_freq, int(use_revcomp), int(merge_metadata))
        rc = self.L.mmt_engine_run_supplied(self.h, cb, None, _p(lens), len(lens), C.byref(p))
        if failure:
            raise failure[0]
        _check(rc)
        return 1

    def run_files(self, paths, out_prefix=None, min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0,
                  use_revcomp=True, merge_metadata=False, max_text_chars=0):
        """FASTA files (one document each) -> PREFIX.mums | .mems + PREFIX.lengths, in-process: the unit of work of
        the reference's build_main.  Returns {"read", "run", "write", "total"} seconds."""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        p = Params(min_match_len, num_distinct, max_doc_freq, max_total_freq, int(use_revcomp), int(merge_metadata))
        sec = (C.c_double * 4)()
        _check(self.L.mmt_engine_run_files(self.h, arr, len(paths), C.byref(p),
                                           os.fsencode(out_prefix) if out_prefix else None, max_text_chars, sec))
        return dict(zip(["read", "run", "write", "total"], map(float, sec)))

    def set_text_sink(self, path):
        """PREFIX.mums / .mems of the next runs straight to `path` while a run goes on (None: off).  A run over a text that fills
        the device keeps nothing else of its rows (mumemto_gpu.h)."""
        _check(self.L.mmt_engine_set_text_sink(self.h, os.fsencode(path) if path else None))

    def release_columns(self, keep_anchor_ranks=False):
        """Text, windows and sort scratch of the last run back to the device heap; keep_anchor_ranks: rows folded later can
        still be put into direct-run order (anchor_merge(sort_like_direct=True))."""
        _check(self.L.mmt_engine_release_columns(self.h, int(keep_anchor_ranks)))

    def merged_thresholds(self, anchor_len):
        out = np.zeros(anchor_len + 1, np.uint16)
        _check(self.L.mmt_copy_merged_thresh(self.h, _p(out)))
        return out

    def set_producer(self, kind="auto", w=0, p=0):
        """kind: 'auto' | 'direct' (reference -g path) | 'pfp' (reference default path) | 'guided' (the parse without the
        suffix array of its dictionary: collections with little redundancy, chosen automatically when needed)."""
        _check(self.L.mmt_engine_set_producer(self.h, {"auto": 0, "direct": 1, "pfp": 2, "guided": 3, "expand": 4}[kind], w, p))

    def producer_used(self):
        return {1: "direct", 2: "pfp", 3: "guided"}.get(self.L.mmt_producer_used(self.h), "?")

    def text_sink_digest(self):
        """(bytes, digest) of what the last run's text sink wrote (sink "/dev/null", or MMT_SINK_DIGEST set)"""
        out = (C.c_uint64 * 2)()
        _check(self.L.mmt_text_sink_digest(self.h, out))
        return int(out[0]), int(out[1])

    def kmer_in_share(self, kmer):
        """True when the suffixes beginning with kmer belong to the share of the stream the last (sharded, bucket-wise) run produced"""
        km = np.frombuffer(bytes(kmer), np.uint8).copy()
        r = self.L.mmt_kmer_in_share(self.h, _p(km), len(km))
        if r < 0:
            raise MumemtoError("the bucket-wise producer did not run")
        return bool(r)

    def set_row_tap(self, kmers, max_rows=1 << 16, max_occ=1 << 24):
        """kmers: list of equal-length byte strings (<= 16 characters), or [] to switch the tap off: the next runs copy every
        accepted interval whose match begins with one of them (length + all its text positions) before its window goes"""
        kmers = [bytes(k) for k in kmers]
        k = len(kmers[0]) if kmers else 0
        assert all(len(x) == k for x in kmers)
        flat = np.frombuffer(b"".join(kmers), np.uint8).copy() if kmers else np.zeros(1, np.uint8)
        _check(self.L.mmt_engine_set_row_tap(self.h, _p(flat), len(kmers), k, max_rows, max_occ))

    def row_tap(self):
        """(length[rows], occ_start[rows + 1], sa[entries]) of the rows the last run tapped, in no particular order"""
        c = (C.c_uint64 * 2)()
        _check(self.L.mmt_row_tap_counts(self.h, c))
        rows, occ = int(c[0]), int(c[1])
        length = np.zeros(max(rows, 1), np.uint32); start = np.zeros(rows + 1, np.uint64); sa = np.zeros(max(occ, 1), np.uint64)
        _check(self.L.mmt_row_tap_get(self.h, _p(length), _p(start), _p(sa)))
        return length[:rows], start, sa[:occ]

    def kmer_positions(self, kmers, cap=1 << 22):
        """(positions ascending, index of the k-mer each begins with) over the text resident on the device"""
        kmers = [bytes(k) for k in kmers]
        k = len(kmers[0])
        flat = np.frombuffer(b"".join(kmers), np.uint8).copy()
        pos = np.zeros(cap, np.uint64); which = np.zeros(cap, np.uint32)
        found = C.c_uint64(0)
        _check(self.L.mmt_kmer_positions(self.h, _p(flat), len(kmers), k, _p(pos), _p(which), cap, C.byref(found)))
        if found.value > cap:
            raise MumemtoError("%d positions begin with those k-mers: more than the %d asked for" % (found.value, cap))
        return pos[:found.value], which[:found.value]

    def producer_stats(self):
        """bucket-wise producer, last run: slices of run bins (assembly gaps), passes over the text, batches, staged"""
        out = (C.c_uint64 * 4)()
        _check(self.L.mmt_producer_stats(self.h, out))
        return {"run_slices": int(out[0]), "text_passes": int(out[1]), "batches": int(out[2]), "staged": bool(out[3])}

    def producer_expanded(self):
        """the bucket-wise producer sorted one representative per (distinct phrase, offset) and the emitter expanded them"""
        return bool(self.L.mmt_producer_expanded(self.h))

    def parse_only(self, use_revcomp=True, w=10, p=100):
        """Text layout + prefix-free parse; returns (dict bytes, parse u32[]) as the reference's -P writes them."""
        _check(self.L.mmt_engine_parse_only(self.h, int(use_revcomp), w, p))
        c = self.pfp_counts()
        d = np.zeros(max(c["dict_len"], 1), np.uint8)
        q = np.zeros(max(c["phrases"], 1), np.uint32)
        _check(self.L.mmt_pfp_copy_dict(self.h, _p(d)))
        _check(self.L.mmt_pfp_copy_parse(self.h, _p(q)))
        return d[: c["dict_len"]].tobytes(), q[: c["phrases"]]

    def pfp_counts(self):
        out = (C.c_uint64 * 8)()
        _check(self.L.mmt_pfp_counts(self.h, out))
        d = dict(zip(["phrases", "distinct", "dict_len", "groups", "rounds_dict", "rounds_parse", "entries",
                      "oversized_groups"], map(int, out)))
        d["run_refined"] = int(self.L.mmt_pfp_run_refined(self.h))
        return d

    def pfp_stage_ms(self):
        out = (C.c_float * 8)()
        _check(self.L.mmt_pfp_stage_ms(self.h, out))
        return list(out)

    # results
    def output_text(self):
        n = C.c_size_t()
        ptr = self.L.mmt_output_text(self.h, C.byref(n))
        return _bytes_at(ptr, n.value)

    def output_size(self):
        """Bytes of the last run's .mums / .mems output; brings them into page-locked host memory (no Python copy)."""
        n = C.c_size_t()
        self.L.mmt_output_text(self.h, C.byref(n))
        return n.value

    def output_bumbl(self):
        n = C.c_size_t()
        ptr = self.L.mmt_output_bumbl(self.h, C.byref(n))
        return _bytes_at(ptr, n.value)

    def rows_mum(self):
        n, N = self.L.mmt_num_rows(self.h), self.L.mmt_num_docs(self.h)
        length = np.zeros(max(n, 1), np.uint32)
        off = np.zeros((max(n, 1), N), np.int64)
        st = np.zeros((max(n, 1), N), np.uint8)
        _check(self.L.mmt_rows_mum(self.h, _p(length), _p(off), _p(st)))
        return length[:n], off[:n], st[:n]

    def rows_mum_device(self):
        """(n_rows, n_docs, length_ptr, offsets_ptr, strands_ptr): the MUM rows of the last run in HBM,
        valid until the next run."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self.L.mmt_rows_mum_device(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return (self.L.mmt_num_rows(self.h), self.L.mmt_num_docs(self.h), a.value or 0, b.value or 0, c.value or 0)

    def rows_mem(self):
        n, t = self.L.mmt_num_rows(self.h), self.L.mmt_num_occ(self.h)
        length = np.zeros(max(n, 1), np.uint32)
        occ = np.zeros(n + 1, np.uint64)
        off = np.zeros(max(t, 1), np.int64)
        ids = np.zeros(max(t, 1), np.uint64)
        st = np.zeros(max(t, 1), np.uint8)
        _check(self.L.mmt_rows_mem(self.h, _p(length), _p(occ), _p(off), _p(ids), _p(st)))
        return length[:n], occ, off[:t], ids[:t], st[:t]

    def thresholds(self):
        n = self.L.mmt_thresh_len(self.h)
        out = np.zeros(max(n, 1), np.uint16)
        if n:
            _check(self.L.mmt_copy_thresh(self.h, _p(out)))
        return out[:n]

    def thresholds32(self):
        """The engine's own threshold column: 32 bits, never saturated (what the fold and the exchange carry)."""
        n = self.L.mmt_thresh_len(self.h)
        out = np.zeros(max(n, 1), np.uint32)
        if n:
            _check(self.L.mmt_copy_thresh32(self.h, _p(out)))
        return out[:n]

    def thresh_device_ptr(self):
        """16-bit form (saturated at 65535 like PREFIX.athresh), made from the 32-bit column on the first call."""
        return self.L.mmt_thresh_device(self.h)

    def thresh_device_ptr32(self):
        return self.L.mmt_thresh_device32(self.h)

    # stage introspection
    def text_length(self):
        return int(self.L.mmt_text_length(self.h))

    def _copy(self, fn, dtype):
        n = self.text_length()
        out = np.zeros(max(n, 1), dtype)
        _check(fn(self.h, _p(out)))
        return out[:n]

    def text(self):
        return self._copy(self.L.mmt_copy_text, np.uint8)

    def sa(self):
        """Suffix array of the last run (uint32 for narrow runs, uint64 when positions need 40 bits)."""
        if self.is_wide():
            return self._copy(self.L.mmt_copy_sa64, np.uint64)
        return self._copy(self.L.mmt_copy_sa, np.uint32)

    def set_scan_shard(self, index, count):
        """This engine produces, scans and drops share `index` of `count` of the stream (multi-GPU runs of the modes
        without a partition merge); the ranks' outputs concatenate to the single-GPU output."""
        _check(self.L.mmt_engine_set_scan_shard(self.h, C.c_uint32(index), C.c_uint32(count)))

    def keep_columns(self, on=True):
        """The columns of the stream exist one window at a time; on=True also copies every window into whole columns, so
        that sa() / lcp() / bwt() work after the run (default: only for texts below 2^26 characters)."""
        _check(self.L.mmt_engine_keep_columns(self.h, C.c_int(1 if on is True else (0 if on is False else int(on)))))

    def columns_kept(self):
        return bool(self.L.mmt_columns_kept(self.h))

    def stream_stats(self):
        """Of the last run: entries of the stream this engine produced, high-water mark of its window buffers (bytes),
        windows, bytes of the suffix-array entries kept with accepted rows."""
        out = (C.c_uint64 * 4)()
        _check(self.L.mmt_stream_stats(self.h, out))
        return {"entries": int(out[0]), "window_bytes": int(out[1]), "windows": int(out[2]), "row_entry_bytes": int(out[3])}

    def sort_pieces(self):
        """[(first suffix-array entry, number of entries)] per rank of the last (sharded) run."""
        k = self.L.mmt_sort_pieces(self.h, None, None, 0)
        first, count = np.zeros(max(k, 1), np.uint64), np.zeros(max(k, 1), np.uint64)
        self.L.mmt_sort_pieces(self.h, _p(first), _p(count), k)
        return [(int(first[i]), int(count[i])) for i in range(k)]

    def is_wide(self):
        return bool(self.L.mmt_is_wide(self.h))

    def scan_ranges(self):
        return int(self.L.mmt_scan_ranges(self.h))

    def device_memory(self):
        """Device heap of this engine's GPU: bytes mapped, live, peak, and seconds spent mapping."""
        out = (C.c_uint64 * 4)()
        _check(self.L.mmt_device_memory(self.h, out))
        return {"mapped": int(out[0]), "live": int(out[1]), "peak": int(out[2]), "map_seconds": out[3] / 1e6}

    def lcp(self):
        return self._copy(self.L.mmt_copy_lcp, np.uint32)

    def bwt(self):
        return self._copy(self.L.mmt_copy_bwt, np.uint8)

    def candidates(self):
        n = self.L.mmt_num_candidates(self.h)
        out = np.zeros((max(n, 1), 4), np.uint32)
        if n:
            _check(self.L.mmt_copy_candidates(self.h, _p(out)))
        return out[:n]

    def stage_ms(self):
        out = (C.c_float * 8)()
        _check(self.L.mmt_stage_ms(self.h, out))
        return list(out)

    def column_bytes(self):
        out = (C.c_uint32 * 3)()
        _check(self.L.mmt_column_bytes(self.h, out))
        return list(out)

    # anchor merge
    def anchor_merge(self, parts, sort_like_direct=False, want_rows=True, min_len=20, text_file=None, slices=0, want_text=True):
        """parts: list of DevicePartition, or of (length u32[n], offsets i64[n,nd], strands u8[n,nd], thresh)
        where thresh is a numpy u16 array (host) or a device address paired as (ptr, length).
        want_rows=False returns only the PREFIX.mums bytes (no D2H of the tables); text_file: the library writes them
        there itself and the result carries no "text"."""
        arr = (Partition * len(parts))()
        keep = []
        for i, part in enumerate(parts):
            if isinstance(part, DevicePartition):
                arr[i] = Partition(part.n_rows, part.n_docs, part.length_ptr, part.offsets_ptr, part.strands_ptr,
                                   part.thresh_ptr, part.thresh_len, 1, 1, part.thresh_bits)
                keep.append(part)
                continue
            length, off, st, th = part
            length = np.ascontiguousarray(length, np.uint32)
            off = np.ascontiguousarray(off, np.int64)
            st = np.ascontiguousarray(st, np.uint8)
            if off.ndim != 2:
                off, st = off.reshape(len(length), -1), st.reshape(len(length), -1)
            if isinstance(th, tuple):       # (device address, entries[, bits])
                tptr, tlen, on_dev = th[0], th[1], 1
                bits = th[2] if len(th) > 2 else 16
            else:                           # uint16: the PREFIX.athresh width; uint32: the engine's own (thresholds32)
                bits = 32 if getattr(th, "dtype", None) == np.uint32 else 16
                th = np.ascontiguousarray(th, np.uint32 if bits == 32 else np.uint16)
                tptr, tlen, on_dev = _p(th).value, len(th), 0
            keep += [length, off, st, th]
            arr[i] = Partition(len(length), off.shape[1], _p(length).value, _p(off).value, _p(st).value, tptr, tlen,
                               on_dev, 0, bits)
        m = C.c_void_p()
        if slices:      # the same table as `slices` independent slices of the anchor (what `slices` ranks fold at once)
            _check(self.L.mmt_anchor_merge_by_ranges(self.h, arr, len(parts), int(slices), C.c_uint32(min_len), C.byref(m)))
        else:
            _check(self.L.mmt_anchor_merge_min_len(self.h, arr, len(parts), C.c_uint32(min_len), C.byref(m)))
        try:
            if sort_like_direct:
                _check(self.L.mmt_merged_sort_like_direct(self.h, m))
            n, nd = self.L.mmt_merged_rows(m), self.L.mmt_merged_docs(m)
            if text_file is not None:
                _check(self.L.mmt_merged_write_text(m, os.fsencode(text_file)))
                text = None
            elif not want_text:
                text = None
            else:
                k = C.c_size_t()
                ptr = self.L.mmt_merged_text(m, C.byref(k))
                if not ptr and n:
                    raise MumemtoError(self.L.mmt_last_error().decode())
                text = _bytes_at(ptr, k.value)
            if not want_rows:
                return dict(text=text, n_rows=n, n_docs=nd)
            length = np.zeros(max(n, 1), np.uint32)
            off = np.zeros((max(n, 1), nd), np.int64)
            st = np.zeros((max(n, 1), nd), np.uint8)
            th = np.zeros(int(arr[0].thresh_len), np.uint16)
            _check(self.L.mmt_merged_get(m, _p(length), _p(off), _p(st), _p(th)))
            return dict(lengths=length[:n], offsets=off[:n], strands=st[:n], thresh=th, text=text, n_rows=n, n_docs=nd)
        finally:
            self.L.mmt_merged_free(m)


def _engine_rows_in_direct_order(self, length, offsets, strands, thresh=None):
    """Rows folded elsewhere (mumemto_amd.dist.fold_by_ranges) -> re-sorted into the order of a direct run by this
    engine's anchor suffix ranks -> PREFIX.mums bytes."""
    length = np.ascontiguousarray(length, np.uint32)
    offsets = np.ascontiguousarray(offsets, np.int64).reshape(len(length), -1)
    strands = np.ascontiguousarray(strands, np.uint8).reshape(len(length), -1)
    th = np.ascontiguousarray(thresh, np.uint16) if thresh is not None else np.zeros(1, np.uint16)
    m = C.c_void_p()
    _check(self.L.mmt_merged_from_rows(self.h, _p(length), _p(offsets), _p(strands), len(length), offsets.shape[1], _p(th),
                                       len(th) if thresh is not None else 0, C.byref(m)))
    try:
        _check(self.L.mmt_merged_sort_like_direct(self.h, m))
        k = C.c_size_t()
        ptr = self.L.mmt_merged_text(m, C.byref(k))
        if not ptr and len(length):
            raise MumemtoError(self.L.mmt_last_error().decode())
        return _bytes_at(ptr, k.value)
    finally:
        self.L.mmt_merged_free(m)


Engine.rows_in_direct_order = _engine_rows_in_direct_order


class Comm:
    """The C-ABI multi-GPU exchange (RCCL, one process per GPU): mmt_comm_* / mmt_dist_* of mumemto_gpu.h.
    Rank 0 calls Comm.unique_id() and hands the 128 bytes to the other ranks out of band; Comm(engine, rank, world, id)
    is collective, and so are merge() and gather_text()."""

    @staticmethod
    def unique_id():
        L = load_library()
        buf = (C.c_uint8 * 128)()
        _check(L.mmt_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, engine, rank, world, unique_id):
        self.L, self.engine, self.rank, self.world = engine.L, engine, rank, world
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _check(self.L.mmt_comm_create(engine.h, rank, world, buf, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mmt_comm_destroy(self.h)
            self.h = None

    def merge(self, min_len=20, by_ranges=False, text_file=None):
        """Strict multi-MUMs: exchange + fold + re-sort.  Rank 0 gets {"text", "n_rows", "n_docs"}, the others None.
        by_ranges: every rank folds its slice of the anchor (automatic from four ranks on; rank 0 decides for everybody).
        text_file: rank 0's library writes PREFIX.mums there itself (in pieces; no Python copy) and "text" is None."""
        m = C.c_void_p()
        f = self.L.mmt_dist_merge_ranges if by_ranges else self.L.mmt_dist_merge
        _check(f(self.h, self.engine.h, C.c_uint32(min_len), C.byref(m)))
        if not m:
            return None
        try:
            n = self.L.mmt_merged_rows(m)
            if text_file is not None:
                _check(self.L.mmt_merged_write_text(m, os.fsencode(text_file)))
                return dict(text=None, n_rows=n, n_docs=self.L.mmt_merged_docs(m))
            k = C.c_size_t()
            ptr = self.L.mmt_merged_text(m, C.byref(k))
            if not ptr and n:
                raise MumemtoError(self.L.mmt_last_error().decode())
            return dict(text=_bytes_at(ptr, k.value), n_rows=n, n_docs=self.L.mmt_merged_docs(m))
        finally:
            self.L.mmt_merged_free(m)

    def loopback(self):
        """The exchange's messages with this rank as its own peer (mmt_comm_loopback): the engine's last run (merge metadata on)
        supplies row tables and thresholds; returns bytes, pieces, the largest piece, mismatching elements, seconds."""
        out = (C.c_uint64 * 8)()
        _check(self.L.mmt_comm_loopback(self.h, out))
        return {"bytes": int(out[0]), "pieces": int(out[1]), "largest_piece_bytes": int(out[2]), "different": int(out[3]),
                "seconds": int(out[4]) / 1e6, "rows": int(out[5]), "cells": int(out[6]), "thresholds": int(out[7])}

    def selftest(self, elements, width=4):
        """One message of that many elements of `width` bytes to this rank itself, in the exchange's pieces (mmt_comm_selftest)."""
        out = (C.c_uint64 * 4)()
        _check(self.L.mmt_comm_selftest(self.h, int(elements), int(width), out))
        return {"different": int(out[0]), "pieces": int(out[1]), "largest_piece_bytes": int(out[2]), "seconds": int(out[3]) / 1e6}

    def gather_text(self):
        """Sharded modes (Engine.set_scan_shard): the whole output on rank 0, b"" elsewhere."""
        ptr, k = C.c_void_p(), C.c_size_t()
        _check(self.L.mmt_dist_gather_text(self.h, C.byref(ptr), C.byref(k)))
        return _bytes_at(ptr.value, k.value) if k.value else b""

