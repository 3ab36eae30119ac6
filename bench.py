#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric: "input Gbp/s end-to-end").

One step = the reference's unit of work, build_main (src/pfp_mum.cpp:31-159): FASTA files (in the page cache) ->
host parse -> H2D -> text layout -> suffix array / LCP / BWT -> LCP-interval match scan -> rows -> PREFIX.mums
written and closed, run in-process through the C ABI (mmt_engine_run_files: the reader and the engine entry
mumemto_exec uses).

N = 1: `value` is SURVEY.md 8(d)'s clock -- every timed step is a FRESH `mumemto_exec` process, timed from process start to
exit (HIP runtime start, first mapping of the device heap, the run, PREFIX.mums closed, process teardown); the same job
in-process on a warm engine is `value_in_process` beside it (what rounds 1 - 4 reported as `value`).  `--in-process` makes
the in-process steps the timed ones again (profilers that follow one process: tests/profile_round*.sh).

N = 1  : workload = BASELINE.json configs[2] stand-in (SURVEY.md 8(d) "C3"): 94 haplotypes x 64 Mbp, per-base
         divergence 0.001, seed 3, strict multi-MUMs -- |T| = 12.03 G characters as ONE suffix array (40-bit
         positions).  Extra keys: the same job as a fresh mumemto_exec process (process start -> exit), the
         HBM-resident engine step (inputs already on the device, output bytes left in page-locked host memory), the
         k_scan roofline and the 1-core CPU oracle on a bounded sample.
N > 1  : the SAME collection split N ways (strong scaling): one rank per GPU (torch.distributed, RCCL), rank r reads
         the anchor + its share of the other haplotypes, runs the single-GPU path with merge metadata, rows and
         thresholds are all-gathered over xGMI, rank 0 folds them (anchor merge), re-sorts into direct-run order and
         writes PREFIX.mums.  value counts every input base once.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
REF_STREAM_BYTES = 11   # the reference's stream record: SA 5 + LCP 5 + BWT 1 (include/common.hpp:59-60)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--haps", type=int, default=94, help="haplotypes of the collection (incl. the anchor)")
    ap.add_argument("--length", type=int, default=64_000_000, help="bases per haplotype")
    ap.add_argument("--divergence", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-sample-bp", type=int, default=2_000_000,
                    help="bases per haplotype given to the 1-core CPU baseline (0 = skip)")
    ap.add_argument("--producer", default="auto", choices=["auto", "direct", "pfp"])
    ap.add_argument("--pfp-w", type=int, default=0)
    ap.add_argument("--pfp-p", type=int, default=0)
    ap.add_argument("--workdir", default=None, help="where the FASTA files and outputs go (default: /dev/shm or $TMPDIR)")
    ap.add_argument("--no-extras", action="store_true", help="skip the process / HBM-resident / CPU legs (N = 1)")
    ap.add_argument("--realistic", action="store_true",
                    help="the timed collection itself carries satellite arrays, microsatellites, assembly gaps, indels and "
                         "inversions (synth.haplotypes_realistic) instead of i.i.d. bases with substitutions")
    ap.add_argument("--check", action="store_true", help="compare the output with the oracle (small sizes only)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N > 1: the C-ABI exchange of dist.cpp (default; mmt_dist_merge: RCCL bound from C++, grouped "
                         "ncclSend / ncclRecv HBM -> HBM, rank 0 folds, from four ranks on every rank folds its slice of the "
                         "anchor) or torch.distributed collectives + Python glue")
    ap.add_argument("--fold", default="rank0", choices=["rank0", "ranges"],
                    help="N > 1 with --exchange torch: rank 0 folds everything (default), or every rank folds its slice of "
                         "the anchor after a slice-wise exchange of the thresholds (SURVEY 8(e), reduce-scatter shape)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device exercise the N > 1 path on a box with one GPU (testing only)")
    ap.add_argument("--share-device", action="store_true", help="every rank uses GPU 0 (testing only)")
    ap.add_argument("--in-process", action="store_true",
                    help="N = 1: time in-process steps on a warm engine (rounds 1 - 4's `value`) instead of fresh mumemto_exec processes")
    ap.add_argument("--pause", type=float, default=6.0,
                    help="N = 1: seconds between two fresh processes, outside the timed sum (a process that starts right behind one "
                         "that gave 120 GB back waits in the driver for that memory to be scrubbed: the previous job's cost)")
    ap.add_argument("--strict-exchange", action="store_true",
                    help="N > 1: a failure of the native exchange (dist.cpp over RCCL) ends the run with rc != 0 instead of "
                         "continuing over torch.distributed's collectives")
    ap.add_argument("--whole-genome", default="auto", choices=["auto", "yes", "no"],
                    help="N = 1: the guarded leg `whole_genome_1gpu` -- BASELINE's second clause, 94 whole-genome haplotypes, as the "
                         "eight rank shares of configs[3] time-multiplexed on this GPU + the fold (tests/big_c4.py in a subprocess "
                         "with a timeout; never able to lose the main line).  auto: when the host has the memory and --no-extras is not set")
    ap.add_argument("--whole-genome-timeout", type=float, default=900.0)
    return ap.parse_args()


def pick_workdir(a, need_bytes):
    if a.workdir:
        os.makedirs(a.workdir, exist_ok=True)
        return tempfile.mkdtemp(prefix="mumemto_bench_", dir=a.workdir), "given"
    for base, kind in (("/dev/shm", "tmpfs"), (tempfile.gettempdir(), "tmp")):
        try:
            if os.path.isdir(base) and shutil.disk_usage(base).free > need_bytes * 1.3:
                return tempfile.mkdtemp(prefix="mumemto_bench_", dir=base), kind
        except OSError:
            pass
    return tempfile.mkdtemp(prefix="mumemto_bench_"), "tmp"


def cpu_baseline(sample):
    """The oracle timed on a bounded sample, one thread: the reference's DEFAULT route -- prefix-free parse (w 10, p 100:
    the reference's defaults), suffix arrays of dictionary and parse, the emitter of pfp_lcp_mum.hpp, the scan -- is the
    baseline; the suffix sort of the whole text (the reference's -g route) is timed beside it.  Both are restatements
    (SA-IS instead of gsacak / sacak_int: those libraries are not in this image), so the kind is "port"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    bp = sum(len(d[0]) for d in sample)
    t0 = time.perf_counter()
    tl, sec, out = O.run_job_timed(sample, pfp=(10, 100))
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    tl2, sec2, out2 = O.run_job_timed(sample)
    dt2 = time.perf_counter() - t0
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": bp / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
            # the reference is single-threaded; the sample is a prefix of every haplotype and the figure stands
            # for the whole collection by linear extrapolation in |T| (the real work grows a little faster than linearly)
            "extrapolated": True, "host_cpu": model, "host_cores_total": os.cpu_count(),
            "sample": "%d haplotypes x first %d bp of the same synthetic pangenome (|T| = %d), strict multi-MUMs, through the "
                      "prefix-free parse (w 10, p 100; the reference's default route): %.1f s of CPU work (parse + "
                      "dictionary / parse suffix arrays + emitter %.1f s, scan+format %.1f s)"
                      % (len(sample), len(sample[0][0]), tl, dt, sec[1], sec[2]),
            "suffix_sort_of_the_whole_text": {"value": bp / dt2 / 1e9, "unit": "Gbp/s",
                                              "note": "the reference's -g route (SA-IS over the text) on the same sample: "
                                                      "%.1f s (sa+lcp+bwt %.1f s); output identical: %s"
                                                      % (dt2, sec2[1], out == out2)}}, out


def whole_genome_leg(a, eng):
    """BASELINE's second clause -- "wall-clock 94 x HPRC whole-genome" -- on ONE GPU: the eight rank shares of configs[3]
    ({anchor + 12 / 11} x 3.05 Gbp, strict multi-MUMs, merge metadata) one after the other, their rows and 32-bit thresholds kept
    on the host, then the fold in eight slices of the anchor + re-sort + the merged rows formatted and copied out
    (tests/big_c4.py --no-file, in a subprocess with a timeout: whatever happens there, the main line is printed).  Needs
    ~200 GB of host memory; skipped -- with the reason -- when the host does not have it."""
    rec = {"workload": "94 haplotypes x 3,050,000,000 bp (divergence 0.001, seed 4), strict multi-MUMs: the 8 rank shares of "
                       "BASELINE configs[3] time-multiplexed on one GPU + fold + re-sort (tests/big_c4.py)"}
    try:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
        for q in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                v = open(q).read().strip()
                if v and v != "max":
                    now = 0
                    try:
                        now = int(open("/sys/fs/cgroup/memory.current").read())
                    except (OSError, ValueError):
                        pass
                    avail = min(avail, (int(v) - now) / 2**30)
                    break
            except (OSError, ValueError):
                pass
        rec["host_available_gb"] = round(avail)
        if avail < 215 and a.whole_genome != "yes":
            rec["skipped"] = "the host has %d GB available, the eight shares' rows and thresholds + one share's bases need ~200" % avail
            return rec
        eng.close()                              # (this process gives its device memory back first)
        eng.L.mmt_pool_trim()
        time.sleep(8.0)
        out_json = os.path.join(tempfile.gettempdir(), "mumemto_wg_%d.json" % os.getpid())
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "big_c4.py"), "--no-file", "--json-out", out_json],
                           capture_output=True, timeout=a.whole_genome_timeout)
        rec["subprocess_s"] = round(time.perf_counter() - t0, 1)
        rec["rc"] = r.returncode
        if r.returncode == 0 and os.path.exists(out_json):
            g = json.load(open(out_json))
            os.unlink(out_json)
            bp = 94 * 3_050_000_000
            device_s = g["shares_run_s"] + g["fold_resort_write_s"]
            rec.update({
                "shares_run_s": g["shares_run_s"], "slowest_share_s": g["slowest_share_s"], "fold_resort_format_s": g["fold_resort_write_s"],
                "merged_rows": g["merged_rows"], "columns": g["columns"], "peak_hbm_gb": g["peak_hbm_gb"],
                "share_run_s": [x["run_s"] for x in g["shares"]], "share_rows": [x["rows"] for x in g["shares"]],
                "one_gpu_wall_clock_s": round(device_s, 1),
                "value": bp / device_s / 1e9, "unit": "Gbp/s",
                "note": "one_gpu_wall_clock_s = the eight shares' passes + fold + re-sort + formatting, one GPU doing eight ranks' work "
                        "one after the other (the generation of the synthetic haplotypes and the host copies between the passes are "
                        "outside it); on eight GPUs the shares run side by side: slowest share + exchange + one slice's fold",
                "projection_8_gpus_s": g["projection_8_gpus_s"],
            })
        else:
            rec["stderr_tail"] = r.stderr.decode(errors="replace")[-600:]
    except subprocess.TimeoutExpired:
        rec["skipped"] = "timeout after %.0f s" % a.whole_genome_timeout
    except Exception as exc:                      # noqa: BLE001 -- this leg must never lose the main line
        rec["skipped"] = "%s: %s" % (type(exc).__name__, exc)
    return rec


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
    if world > 1 and "MUMEMTO_READ_THREADS" not in os.environ:
        # the ranks of one node share the node's CPUs (and a container's CPU quota): each reader takes its share
        cpus = os.cpu_count() or 1
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cpus = min(cpus, max(1, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
        os.environ["MUMEMTO_READ_THREADS"] = str(max(2, cpus // world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(a.backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    # ---- the collection as FASTA files in the page cache: this rank's documents only -------------------------------
    groups = mdist.partition_docs(a.haps, world)
    mine = groups[rank]
    workdir, work_kind = pick_workdir(a, len(mine) * a.length * 1.02)
    paths, sample = [], []
    t_gen = time.perf_counter()
    if a.realistic:
        a.no_extras = True                      # (the extra legs assume haplotypes of equal length)
    gen = synth.haplotypes_realistic if a.realistic else synth.haplotypes_sparse
    for h, bases in gen(a.haps, a.length, a.divergence, a.seed, which=mine):
        p = os.path.join(workdir, "hap%03d.fa" % h)
        synth.write_fasta_fast(p, bases, name="hap%03d" % h)
        paths.append(p)
        if world == 1 and a.cpu_sample_bp > 0 and not a.no_extras:
            sample.append([bases[: a.cpu_sample_bp].tobytes()])
    t_gen = time.perf_counter() - t_gen
    out_prefix = os.path.join(workdir, "out")

    # ---- N = 1: the timed steps are FRESH processes -- mumemto_exec, process start -> exit (HIP runtime start, first mapping
    #      of the device heap, the run, PREFIX.mums closed, teardown) -- run before this process has touched the device.
    #      "wall-clock from process start to last byte of PREFIX.mums closed".
    exe = os.path.join(ROOT, "mumemto_amd", "bin", "mumemto_exec")
    fresh = None
    if rank == 0 and world == 1 and not a.in_process and not a.realistic and os.path.exists(exe):
        stats = os.path.join(workdir, "cli_stats.json")
        runs = []
        for i in range(a.warmup + a.steps):
            if i:
                time.sleep(a.pause)             # (outside the sum: see --pause)
            if os.path.exists(stats):
                os.unlink(stats)
            t0 = time.perf_counter()
            r = subprocess.run([exe] + paths + ["-o", os.path.join(workdir, "cli")], capture_output=True,
                               env=dict(os.environ, MUMEMTO_STATS=stats, MUMEMTO_DEVICE=str(local_rank)))
            wall = time.perf_counter() - t0
            rec = {"wall_s": wall, "rc": r.returncode, "timed": i >= a.warmup}
            if r.returncode == 0 and os.path.exists(stats):
                st = json.load(open(stats))
                rec.update({"heap_map_seconds": st["heap_map_seconds"], "heap_peak_bytes": st["heap_peak_bytes"],
                            "stage_ms": st["stage_ms"], "text_chars": st["text_chars"], "scan_ranges": st["scan_ranges"],
                            "rows": st["rows"], "candidates": st["candidates"], "wide": st["wide"],
                            "seconds_to_outputs_written": st["seconds_since_start"]})
            else:
                rec["stderr_tail"] = r.stderr.decode(errors="replace")[-400:]
            runs.append(rec)
        timed = [x for x in runs if x["timed"]]
        # the same job with NO pause between the processes: what a process that starts right behind another one pays for the
        # memory its predecessor gave back (the driver scrubs it; the figure the pause keeps out of `value`)
        b2b = []
        if timed and all(x["rc"] == 0 for x in timed) and not a.no_extras:
            for i in range(4):
                t0 = time.perf_counter()
                r = subprocess.run([exe] + paths + ["-o", os.path.join(workdir, "cli")], capture_output=True,
                                   env=dict(os.environ, MUMEMTO_STATS=stats, MUMEMTO_DEVICE=str(local_rank)))
                b2b.append({"wall_s": time.perf_counter() - t0, "rc": r.returncode})
            time.sleep(a.pause)
        if timed and all(x["rc"] == 0 and "stage_ms" in x for x in timed):
            fresh = {"runs": runs, "timed": timed, "seconds": sum(x["wall_s"] for x in timed), "back_to_back": b2b}
        else:
            sys.stderr.write("[bench] the fresh-process steps failed (%s): the in-process steps are the timed ones\n"
                             % [x.get("stderr_tail", x["rc"]) for x in runs if x["rc"] != 0][:1])
    cli = None

    eng = mumemto_amd.Engine(local_rank)
    eng.set_producer(a.producer, a.pfp_w, a.pfp_p)
    merge_mode = world > 1
    L0 = a.length
    comm = None
    if merge_mode and a.exchange == "native":
        box = [mumemto_amd.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)          # the 128-byte id travels out of band (here: torch's store)
        comm = mumemto_amd.Comm(eng, rank, world, box[0])
    phases = {"read": 0.0, "run": 0.0, "write": 0.0, "exchange_fold": 0.0}
    state = {"comm": comm, "exchange": a.exchange if merge_mode else None}

    def step(timed):
        if not merge_mode:
            sec = eng.run_files(paths, out_prefix=out_prefix)
            if timed:
                for k in ("read", "run", "write"):
                    phases[k] += sec[k]
            return None
        sec = eng.run_files(paths, out_prefix=None, merge_metadata=True)
        t0 = time.perf_counter()
        if state["comm"] is not None:   # C-ABI exchange: HBM -> HBM sends, fold and re-sort, PREFIX.mums written by the library
            merged = state["comm"].merge(min_len=20, text_file=out_prefix + ".mums")
            if timed:
                phases["read"] += sec["read"]; phases["run"] += sec["run"]
                phases["exchange_fold"] += time.perf_counter() - t0
            return merged
        if a.fold == "ranges":          # every rank folds its slice of the anchor (thresholds travel slice-wise)
            def fold(parts):
                m = eng.anchor_merge(parts)
                return m["lengths"], m["offsets"], m["strands"], m["thresh"]
            lr, orows, srows = eng.rows_mum()
            ml, mo, ms = mdist.merge_by_ranges(fold, (lr, orows, srows, eng.thresholds()[: L0 + 1]), dist,
                                               torch.device("cpu") if a.backend == "gloo" else device, L0 + 1)
            merged = None
            if rank == 0:
                merged = {"text": eng.rows_in_direct_order(ml, mo, ms)}
                with open(out_prefix + ".mums", "wb") as f:
                    f.write(merged["text"])
            if timed:
                phases["read"] += sec["read"]; phases["run"] += sec["run"]
                phases["exchange_fold"] += time.perf_counter() - t0
            return merged
        # rows and thresholds go from this rank's HBM straight into the all-gather; rank 0 folds them in HBM
        len_t, off_t, st_t = mdist.engine_rows_as_tensors(eng, device)
        th = torch.as_tensor(mdist.DevicePointerView(eng.thresh_device_ptr(), L0 + 1), device=device)
        parts = mdist.all_gather_partitions_device((len_t, off_t, st_t, th), dist)
        merged = None
        if rank == 0:
            merged = eng.anchor_merge(mdist.device_partitions(parts), sort_like_direct=True, want_rows=False,
                                      text_file=out_prefix + ".mums")
        if timed:
            phases["read"] += sec["read"]; phases["run"] += sec["run"]
            phases["exchange_fold"] += time.perf_counter() - t0
        return merged

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    if state["comm"] is not None:
        # The native exchange (dist.cpp: ncclSend / ncclRecv groups over the communicator above) has met real RCCL between two
        # GPUs nowhere yet -- the boxes this was built on have one; its glue runs over a test double.  One untimed step tries
        # it; if ANY rank fails there, every rank says so on stderr and the run goes on over torch.distributed's collectives
        # (the same RCCL, the other route of this file), with the reason in the result line.  Nothing is hidden, nothing
        # leaves the GPUs, and the timed steps all take one route.
        ok, why = 1, ""
        try:
            step(False)
        except Exception as exc:                      # noqa: BLE001 (whatever the transport raised)
            ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and a.strict_exchange:
            sys.stderr.write("[bench rank %d] native exchange failed in the trial step (%s): --strict-exchange, giving up\n"
                             % (rank, why or "on another rank"))
            dist.barrier()
            sys.exit(3)
        if int(flag.item()) == 0:
            sys.stderr.write("[bench rank %d] native exchange failed in the trial step (%s): continuing over torch.distributed\n"
                             % (rank, why or "on another rank"))
            state["comm"] = None
            state["exchange"] = "torch.distributed (the native exchange failed in the trial step%s)" % (": " + why if why else "")
    # (with fresh processes as the timed steps, the in-process leg -- `value_in_process`, stage and phase averages -- is one
    # warm-up + two steps whatever --steps says)
    in_steps, in_warmup = (a.steps, a.warmup) if fresh is None else (2, 1)
    for _ in range(in_warmup):
        step(False)
    scan_ms, stage_acc = [], np.zeros(8)
    fence()
    t0 = time.perf_counter()
    for _ in range(in_steps):
        step(True)
        ms = eng.stage_ms()
        scan_ms.append(ms[3])
        stage_acc += np.array(ms)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_bp = a.length * a.haps                # every distinct input base once
    n_text = eng.text_length()
    col = eng.column_bytes()
    algo_bytes = float(sum(col)) * n_text       # SA + LCP + BWT columns of the stream as stored, one pass
    scan_avg_ms = float(np.mean(scan_ms))
    if fresh is not None:                        # the scan kernel's HIP-event time inside the TIMED processes (MUMEMTO_STATS)
        scan_avg_ms = float(np.mean([x["stage_ms"][3] for x in fresh["timed"]]))
    scan_avg_ms = max(scan_avg_ms, 1e-9)         # (a crippled timing run -- MMT_EMIT_ABLATE -- scans nothing)
    value_in_process = total_bp * in_steps / dt / 1e9
    ms_in_process = dt / in_steps * 1e3
    if fresh is not None:
        value, ms_per_step = total_bp * a.steps / fresh["seconds"] / 1e9, fresh["seconds"] / a.steps * 1e3
    else:
        value, ms_per_step = value_in_process, ms_in_process
    achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9
    out_file = out_prefix + ".mums"
    out_bytes = os.path.getsize(out_file) if rank == 0 and os.path.exists(out_file) else 0
    result = {
        "metric": "input Gbp/s end-to-end",
        "value": value,
        "unit": "Gbp/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8 text, u32 (+u8 beyond 2^32) positions, u32 LCP",
        "data": "synthetic",
        "config": {
            "workload": "%d haplotypes x %d bp synthetic pangenome (divergence %g, seed %d) as FASTA in the page cache "
                        "(%s), strict multi-MUMs (-l 20, revcomp on) -> PREFIX.mums on disk; %s"
                        % (a.haps, a.length, a.divergence, a.seed, work_kind,
                           "BASELINE configs[2] stand-in (94 HPRC haplotypes, chr20)" if (a.haps, a.length) == (94, 64_000_000)
                           else "scaled variant of the BASELINE configs[2] stand-in"),
            "haplotypes": a.haps, "bases_per_haplotype": a.length, "text_chars_per_gpu": int(n_text),
            "one_suffix_array": bool(eng.L.mmt_partitions_used(eng.h) == 1), "positions_40_bit": eng.is_wide(),
            "scan_ranges": eng.scan_ranges(),
            "stream": "produced, scanned and dropped window by window: %d windows, %.2f GB of window buffers (the suffix "
                      "array / BWT / LCP columns are never stored as a whole)" % (
                          eng.stream_stats()["windows"], eng.stream_stats()["window_bytes"] / 1e9),
            "parallelism": "1 GPU" if world == 1 else "anchor partitions x%d + RCCL %s + GPU fold" % (
                world, "sends through the C ABI (mmt_dist_merge, dist.cpp)" if state["comm"] is not None else "all-gather (torch.distributed)"),
            "timed_region": ("every step a FRESH mumemto_exec process, process start -> exit (SURVEY.md 8(d)): HIP runtime start, "
                             "FASTA files (page cache) -> host parse -> H2D -> GPU path -> PREFIX.mums closed, process teardown; "
                             "%.0f s pause between two processes outside the sum (the driver scrubs what the process before gave "
                             "back: the previous job's cost)" % a.pause) if fresh is not None else
                            "FASTA files (page cache) -> host parse -> H2D -> GPU path -> PREFIX.mums closed; in-process "
                            "(HIP runtime up, device heap mapped by the warm-up step)",
            "output_bytes": out_bytes, "output_rows": int(eng.L.mmt_num_rows(eng.h)) if world == 1 else None,
            "scan_candidates": int(eng.L.mmt_num_candidates(eng.h)),
            "stream_producer": eng.producer_used(),
            "exchange": state["exchange"],
            "exchange_detail": None if world == 1 else {
                "world": world, "route": ("native: mmt_dist_merge (dist.cpp), %s" % ("every rank folds its slice of the anchor after an "
                                          "all-to-all of row and threshold slices" if world >= 4 else "rank 0 folds what the others send"))
                if state["comm"] is not None else "torch.distributed all-gather + fold on rank 0",
                "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if a.backend == "nccl" else None,
                "strict": bool(a.strict_exchange),
                "bytes_sent_per_rank_estimate": int((L0 + 1) * 4 + int(eng.L.mmt_num_rows(eng.h)) * (4 + 9 * len(mine))),
                # every message of dist.cpp travels in pieces of at most 2^29 bytes (dist.cpp rccl_chunk_elements: pieces beyond
                # 1 GiB lost half their elements in the real library, profiles/round6_rccl_piece_sizes.log)
                "messages_of_this_rank": [
                    {"what": what, "elements": int(cnt), "bytes": int(cnt * width), "pieces": int(max(1, -(-cnt // ((1 << 29) // width)))),
                     "largest_piece_bytes": int(min(cnt, (1 << 29) // width) * width)}
                    for what, cnt, width in (("thresholds over the anchor (u32)", L0 + 1, 4),
                                             ("row lengths (u32)", int(eng.L.mmt_num_rows(eng.h)), 4),
                                             ("row offsets (i64)", int(eng.L.mmt_num_rows(eng.h)) * len(mine), 8),
                                             ("row strands (u8)", int(eng.L.mmt_num_rows(eng.h)) * len(mine), 1))],
            },
        },
        "phase_s_avg": {k: v / in_steps for k, v in phases.items()},
        "roofline": {"bound": "hbm", "kernel": "k_scan (LCP-interval match scan), %d launches per step" % eng.scan_ranges(),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None,
                     "algorithmic_bytes_per_suffix": sum(col), "suffixes_per_step": int(n_text),
                     "kernel_ms_per_step": scan_avg_ms,
                     "frac_at_reference_widths": REF_STREAM_BYTES * n_text / (scan_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "stage_ms_avg": {k: float(v) / in_steps for k, v in zip(
            ["text", "suffix_sort", "lcp_bwt", "scan_kernel", "verify", "rows", "stream_windows_in_suffix_sort", "engine_total"],
            stage_acc)},
        # the largest kernel of the step is not the roofline kernel: the emitter writes the windows of the stream (SA 5 B on a
        # wide text + BWT 1 + LCP 4 per suffix) and is bound by latency and VALU work, not by HBM
        "largest_kernel": {"kernel": "stream windows (k_emit2: expands the phrase-suffix groups into SA / BWT / LCP entries)",
                           "ms_per_step": float(stage_acc[6]) / in_steps, "bytes_written_per_suffix": sum(col),
                           "achieved": sum(col) * n_text / max(float(stage_acc[6]) / in_steps, 1e-9) / 1e6, "unit": "GB/s",
                           "frac": sum(col) * n_text / max(float(stage_acc[6]) / in_steps, 1e-9) / 1e6 / HBM_PEAK_GBS},
        "device_memory": eng.device_memory(),
        "generate_s": t_gen,
    }
    # HBM bytes the scan kernel really moved, from the PMC passes under profiles/ (separate rocprofv3 runs of this
    # command line; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), valid for the default workload only
    def newest(name):
        for r in ("round6", "round5", "round4", "round3"):
            f = os.path.join(ROOT, "profiles", "%s_%s" % (r, name))
            if os.path.exists(f):
                return f
        return os.path.join(ROOT, "profiles", "round4_" + name)
    pmc_file = newest("scan_pmc.json")
    if world == 1 and os.path.exists(pmc_file) and (a.haps, a.length, a.divergence, a.seed) == (94, 64_000_000, 0.001, 3):
        pmc = json.load(open(pmc_file))
        result["roofline"]["traffic"] = pmc["hbm_bytes_per_step"]
        result["roofline"]["traffic_unit"] = ("bytes per step = all k_scan launches of one pass (PMC, profiles/%s: %s)"
                                              % (os.path.basename(pmc_file), pmc.get("kernel", "k_scan")))
        result["roofline"]["frac_moved"] = pmc["hbm_bytes_per_step"] / (scan_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    emit_pmc = newest("emit_pmc.json")
    if world == 1 and os.path.exists(emit_pmc) and (a.haps, a.length, a.divergence, a.seed) == (94, 64_000_000, 0.001, 3):
        # the largest kernel's HBM bytes from its own PMC passes: the ratio to its algorithmic bytes says how much it re-reads
        pmc = json.load(open(emit_pmc))
        lk = result["largest_kernel"]
        lk["traffic"] = pmc["hbm_bytes_per_step"]
        lk["traffic_upper_bound"] = pmc.get("hbm_bytes_per_step_fetch_x2")
        lk["algorithmic_bytes_per_step"] = pmc["algorithmic_bytes_per_step"]
        lk["traffic_ratio"] = pmc["hbm_bytes_per_step"] / pmc["algorithmic_bytes_per_step"]
        lk["frac_moved"] = pmc["hbm_bytes_per_step"] / max(lk["ms_per_step"], 1e-9) / 1e6 / HBM_PEAK_GBS
        lk["traffic_unit"] = ("bytes per step = all %s launches of one pass (PMC, profiles/%s: FETCH_SIZE as "
                              "reported -- 8 / 4 bytes per lane gathers, a width the guide's x2 is not calibrated for; the x2 "
                              "figure is traffic_upper_bound -- + WRITE_SIZE)" % (pmc.get("kernel", "k_emit").split("<")[0].split("::")[-1],
                                                                                os.path.basename(emit_pmc)))
    if eng.producer_used() == "pfp":
        result["pfp"] = {"counts": eng.pfp_counts(), "last_step_ms": dict(zip(
            ["triggers_phrases", "distinct_phrases", "dictionary_text", "dictionary_sa", "dictionary_groups",
             "parse_sa", "lists_emit", "total_host_clock"], [round(x, 3) for x in eng.pfp_stage_ms()]))}

    if fresh is not None:
        result["value_in_process"] = value_in_process          # rounds 1 - 4's `value`: warm engine, HIP runtime up, heap mapped
        result["ms_per_step_in_process"] = ms_in_process
        result["value_process_start"] = value                   # (the same figure under the name earlier rounds used)
        result["fresh_processes"] = {
            "runs": [{k: v for k, v in x.items() if k in ("wall_s", "rc", "timed", "heap_map_seconds", "seconds_to_outputs_written")}
                     for x in fresh["runs"]],
            "stage_ms_of_the_timed_runs": [x["stage_ms"] for x in fresh["timed"]],
            "heap_peak_bytes": fresh["timed"][-1]["heap_peak_bytes"],
            "back_to_back": None if not fresh.get("back_to_back") or any(x["rc"] for x in fresh["back_to_back"]) else {
                "runs_wall_s": [round(x["wall_s"], 3) for x in fresh["back_to_back"]],
                "ms_per_step": float(np.mean([x["wall_s"] for x in fresh["back_to_back"][1:]])) * 1e3,
                "value": total_bp / float(np.mean([x["wall_s"] for x in fresh["back_to_back"][1:]])) / 1e9, "unit": "Gbp/s",
                "deferred_release_s": float(np.mean([x["wall_s"] for x in fresh["back_to_back"][1:]])) - fresh["seconds"] / a.steps,
                "note": "processes started one right behind the other (the first of the four follows the pause and is left out of the "
                        "mean): deferred_release_s is what a process waits for the device memory its predecessor released"},
            "output_identical_to_in_process": subprocess.run(["cmp", "-s", os.path.join(workdir, "cli.mums"), out_file]).returncode == 0,
        }
    else:
        result["value_in_process"] = value
        result["value_process_start"] = None
    if rank == 0 and os.path.exists(out_file):
        import hashlib
        h = hashlib.sha256()
        with open(out_file, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        result["config"]["output_sha256_16"] = h.hexdigest()[:16]
        fx_path = os.path.join(ROOT, "tests", "golden", "c3_standin_output.json")
        if world == 1 and not a.realistic and (a.haps, a.length, a.divergence, a.seed) == (94, 64_000_000, 0.001, 3) and os.path.exists(fx_path):
            # the bytes tests/test_gpu_fullsize.py::test_c3_standin_at_full_size_one_suffix_array checks (-m gpu)
            fx = json.load(open(fx_path))
            result["config"]["output_is_the_checked_one"] = bool(fx["output_sha256_16"] == h.hexdigest()[:16] and
                                                                   fx["output_bytes"] == out_bytes)
    if rank == 0 and world == 1 and not a.no_extras and not a.realistic:
        # the same collection shape with the content real assemblies carry (satellite arrays, microsatellites, assembly
        # gaps, indels, inversions: synth.haplotypes_realistic) through the same timed region, one warm-up + one step
        rdir = os.path.join(workdir, "realistic")
        os.makedirs(rdir, exist_ok=True)
        rpaths, rbp = [], 0
        for h, bases in synth.haplotypes_realistic(a.haps, a.length, a.divergence, a.seed):
            p = os.path.join(rdir, "hap%03d.fa" % h)
            synth.write_fasta_fast(p, bases, name="hap%03d" % h)
            rpaths.append(p)
            rbp += len(bases)
        rt = []
        for i in range(2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            eng.run_files(rpaths, out_prefix=os.path.join(rdir, "out"))
            rt.append(time.perf_counter() - t0)
        result["realistic"] = {"ms_per_step": rt[1] * 1e3, "value": rbp / rt[1] / 1e9, "unit": "Gbp/s",
                               "ratio_to_the_iid_step": rt[1] * 1e3 / ms_in_process,
                               "content": "two satellite arrays (period 171, 2.3 % + 3.9 % of the length, 1.5 % diverged "
                                          "copies), twenty microsatellites (period 2 - 6, 10 - 100 kbp), three runs of N "
                                          "(50 kbp, 200 kbp, 1 Mbp), indels 1e-4 per base, an inversion in every seventh "
                                          "haplotype; substitutions as in the i.i.d. collection",
                               "output_rows": int(eng.L.mmt_num_rows(eng.h)), "stage_ms": [round(x, 2) for x in eng.stage_ms()],
                               "pfp_counts": eng.pfp_counts() if eng.producer_used() == "pfp" else None}
        shutil.rmtree(rdir, ignore_errors=True)
    if rank == 0 and world == 1 and not a.no_extras:
        # HBM-resident engine step: bases on the device before the timed region, output bytes in page-locked host
        #     memory after it (what round 1 reported as `value`)
        flat = np.empty(a.haps * a.length, np.uint8)
        for h, bases in synth.haplotypes_sparse(a.haps, a.length, a.divergence, a.seed):
            flat[h * a.length:(h + 1) * a.length] = bases
        d_bases = torch.from_numpy(flat).to(device)
        del flat
        lens = np.full(a.haps, a.length, np.uint64)
        eng.set_input_device(d_bases.data_ptr(), lens, keepalive=d_bases)
        hb = []
        for i in range(3):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            eng.run(min_match_len=20, num_distinct=0, max_doc_freq=1, max_total_freq=0, use_revcomp=True)
            eng.output_size()
            hb.append(time.perf_counter() - t0)
        result["hbm_resident"] = {"ms_per_step": min(hb[1:]) * 1e3, "value": total_bp / min(hb[1:]) / 1e9, "unit": "Gbp/s",
                                  "first_step_ms": hb[0] * 1e3, "stage_ms": [round(x, 2) for x in eng.stage_ms()]}
    if rank == 0 and world == 1 and a.whole_genome != "no" and (a.whole_genome == "yes" or not a.no_extras):
        result["whole_genome_1gpu"] = whole_genome_leg(a, eng)
    if rank == 0:
        if world == 1 and sample and not a.no_extras and a.cpu_sample_bp > 0:
            cb, cpu_out = cpu_baseline(sample)
            result["cpu_baseline"] = cb
            if a.cpu_sample_bp >= a.length:
                result["config"]["output_equals_cpu_oracle"] = bool(cpu_out == open(out_file, "rb").read())
        else:
            result["cpu_baseline"] = None
        if a.check and not (world == 1 and a.cpu_sample_bp >= a.length and not a.no_extras):
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as O
            every = [[b.tobytes()] for _, b in synth.haplotypes_sparse(a.haps, a.length, a.divergence, a.seed)]
            order = mdist.merged_column_order(groups)
            result["config"]["output_equals_cpu_oracle"] = bool(
                O.run([every[i] for i in order]).text() == open(out_file, "rb").read())
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
