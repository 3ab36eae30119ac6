#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric:
"input Gbp/s end-to-end").

One step = one pass of the hot path (text layout -> suffix array / LCP / BWT ->
LCP-interval match scan -> rows -> .mums bytes) over one synthetic pangenome
that is already resident in HBM when the timed region starts.

N = 1  : workload = BASELINE.json configs[1] stand-in: 16 haplotypes x 12.1 Mbp,
         per-base divergence 0.005, seed 2 (SURVEY.md 8(d) "C2"), strict multi-MUMs.
N > 1  : one rank per GPU (torch.distributed, RCCL).  Rank r processes
         {anchor} + its own 15 haplotypes (per-GPU work fixed => weak scaling);
         candidate rows + thresholds are all-gathered and rank 0 folds them
         (anchor merge) and re-sorts into direct-run order.  value counts every
         distinct input base once.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--haps", type=int, default=16, help="haplotypes per GPU (incl. the anchor)")
    ap.add_argument("--length", type=int, default=12_100_000, help="bases per haplotype")
    ap.add_argument("--divergence", type=float, default=0.005)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-sample-bp", type=int, default=3_000_000,
                    help="bases per haplotype given to the 1-core CPU baseline (0 = skip)")
    ap.add_argument("--producer", default="auto", choices=["auto", "direct", "pfp"])
    ap.add_argument("--pfp-w", type=int, default=0)
    ap.add_argument("--pfp-p", type=int, default=0)
    ap.add_argument("--merge-metadata", action="store_true", help="record anchor thresholds also on 1 GPU")
    ap.add_argument("--check", action="store_true", help="compare the output with the oracle (small sizes only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device exercise the N > 1 path on a box with one GPU (testing only)")
    ap.add_argument("--share-device", action="store_true", help="every rank uses GPU 0 (testing only)")
    return ap.parse_args()


def cpu_baseline(docs, sample_bp):
    """The oracle (CPU restatement of the reference's -g path + scan, 1 thread)
    timed on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    sample = [[d[0][:sample_bp]] for d in docs]
    bp = sum(len(d[0]) for d in sample)
    t0 = time.perf_counter()
    tl, sec, out = O.run_job_timed(sample)
    dt = time.perf_counter() - t0
    return {"value": bp / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": "%d haplotypes x first %d bp of the same synthetic pangenome (|T| = %d), strict multi-MUMs; "
                      "%.1f s of CPU work (sa+lcp+bwt %.1f s, scan+format %.1f s)"
                      % (len(sample), sample_bp, tl, dt, sec[1], sec[2])}, out, sample


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import mumemto_amd
    from mumemto_amd import dist as mdist
    from mumemto_amd import synth

    if a.share_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(a.backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    # ---- synthetic pangenome: anchor + (haps-1) haplotypes per rank -----------------------------
    n_total_haps = 1 + world * (a.haps - 1)
    groups = mdist.partition_docs(n_total_haps, world)
    mine = groups[rank]
    all_docs = synth.pangenome(n_total_haps, a.length, a.divergence, a.seed) if world == 1 else None
    if world == 1:
        docs = all_docs
    else:  # every rank generates only what it needs (same generator, same seeds)
        docs = synth.pangenome_subset(n_total_haps, a.length, a.divergence, a.seed, mine)
    doc_len = np.array([len(d[0]) for d in docs], np.uint64)
    flat = np.frombuffer(b"".join(d[0] for d in docs), np.uint8)
    d_bases = torch.from_numpy(flat.copy()).to(device)          # inputs resident in HBM
    stream = torch.cuda.current_stream(device)
    eng = mumemto_amd.Engine(local_rank, stream.cuda_stream)
    eng.set_input_device(d_bases.data_ptr(), doc_len, keepalive=d_bases)
    eng.set_producer(a.producer, a.pfp_w, a.pfp_p)
    merge_mode = world > 1
    want_thresh = merge_mode or a.merge_metadata
    L0 = int(doc_len[0])

    def step():
        eng.run(min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, use_revcomp=True,
                merge_metadata=want_thresh)
        if not merge_mode:
            return eng.output_size()      # the .mums bytes are in (page-locked) host memory at this point
        # rows and thresholds go from this rank's HBM straight into the all-gather; rank 0 folds them in HBM
        len_t, off_t, st_t = mdist.engine_rows_as_tensors(eng, device)
        th = torch.as_tensor(mdist.DevicePointerView(eng.thresh_device_ptr(), L0 + 1), device=device)
        parts = mdist.all_gather_partitions_device((len_t, off_t, st_t, th), dist)
        if rank != 0:
            return b""
        merged = eng.anchor_merge(mdist.device_partitions(parts), sort_like_direct=True, want_rows=False)
        return merged["text"]

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(a.warmup):
        out = step()
    scan_ms, stage_acc = [], np.zeros(8)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
        ms = eng.stage_ms()
        scan_ms.append(ms[3])
        stage_acc += np.array(ms)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not merge_mode:
        out = eng.output_text()

    total_bp = a.length * n_total_haps          # every distinct input base once
    n_text = eng.text_length()
    col = eng.column_bytes()
    algo_bytes = float(sum(col)) * n_text       # SA + LCP + BWT columns of the stream, one pass
    scan_avg_ms = float(np.mean(scan_ms))
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9
    result = {
        "metric": "input Gbp/s end-to-end",
        "value": total_bp * a.steps / dt / 1e9,
        "unit": "Gbp/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/u32",
        "data": "synthetic",
        "config": {"workload": "%d haplotypes x %d bp synthetic pangenome (divergence %g, seed %d), strict multi-MUMs "
                               "(-l 20, revcomp on); BASELINE configs[1] stand-in (16 S. cerevisiae ~12 Mbp)"
                               % (n_total_haps, a.length, a.divergence, a.seed),
                   "haplotypes": n_total_haps, "bases_per_haplotype": a.length, "text_chars_per_gpu": int(n_text),
                   "parallelism": "1 GPU" if world == 1 else "anchor partitions x%d + RCCL all-gather + GPU fold" % world,
                   "output_bytes": len(out), "output_rows": out.count(b"\n"),
                   "scan_candidates": int(eng.L.mmt_num_candidates(eng.h))},
        "roofline": {"bound": "hbm", "kernel": "k_scan (LCP-interval match scan)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None,
                     "algorithmic_bytes_per_suffix": sum(col), "suffixes_per_launch": int(n_text),
                     "avg_kernel_ms": scan_avg_ms},
        "stage_ms_avg": {k: float(v) / a.steps for k, v in zip(
            ["text", "suffix_sort", "lcp_bwt", "scan_kernel", "verify", "rows_gather_d2h", "host_rows_format",
             "engine_total"], stage_acc)},
    }
    # HBM traffic of the scan kernel from the PMC passes of profiles/ (separate rocprofv3 runs of this very
    # command line; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), valid for this workload only
    pmc_file = os.path.join(ROOT, "profiles", "round1_l_scan_pmc.json")
    if world == 1 and os.path.exists(pmc_file) and (a.haps, a.length, a.divergence, a.seed) == (16, 12_100_000, 0.005, 2):
        result["roofline"]["traffic"] = json.load(open(pmc_file))["hbm_bytes_per_launch"]
        result["roofline"]["traffic_unit"] = "bytes per launch (PMC, profiles/round1_l_scan_pmc.json)"
    result["config"]["stream_producer"] = eng.producer_used()
    if eng.producer_used() == "pfp":
        result["pfp"] = {"counts": eng.pfp_counts(), "last_step_ms": dict(zip(
            ["triggers_phrases", "distinct_phrases", "dictionary_text", "dictionary_sa", "dictionary_lcp_groups",
             "parse_sa", "text_keys_sort", "total_host_clock"], [round(x, 3) for x in eng.pfp_stage_ms()]))}
    if rank == 0:
        if world == 1 and a.cpu_sample_bp > 0:
            cb, cpu_out, sample = cpu_baseline(docs, min(a.cpu_sample_bp, a.length))
            result["cpu_baseline"] = cb
            if a.check or min(a.cpu_sample_bp, a.length) == a.length:
                result["config"]["output_equals_cpu_oracle"] = bool(cpu_out == out)
        else:
            result["cpu_baseline"] = None
            if a.check:      # N > 1: the merged output must be the direct run's on the union, in fold column order
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import pyoracle as O
                every = synth.pangenome(n_total_haps, a.length, a.divergence, a.seed)
                order = mdist.merged_column_order(groups)
                result["config"]["output_equals_cpu_oracle"] = bool(O.run([every[i] for i in order]).text() == out)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
