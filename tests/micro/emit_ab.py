"""GPU box helper: the bench workload (or --realistic content) through mumemto_exec once per VARIANT of the environment,
the collection generated once.  Prints per variant: wall clock, GPU stage times (MUMEMTO_STATS), sha256 of PREFIX.mums.
usage: emit_ab.py [--reps N] [--realistic] [--haps H --length L] NAME=ENV1=V1,ENV2=V2 ...   (NAME= alone: no variables)"""
import argparse, hashlib, json, os, shutil, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mumemto_amd import synth, build

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--realistic", action="store_true")
ap.add_argument("--haps", type=int, default=94)
ap.add_argument("--length", type=int, default=64_000_000)
ap.add_argument("--divergence", type=float, default=0.001)
ap.add_argument("--seed", type=int, default=3)
ap.add_argument("--args", default="", help="extra arguments of mumemto_exec, space separated")
ap.add_argument("--pause", type=float, default=7.0)
ap.add_argument("--grouped", action="store_true", help="all repetitions of a variant one after the other (default: the variants take turns)")
ap.add_argument("--stderr", default="", help="print the lines of mumemto_exec's stderr that contain this text (e.g. '[sort]')")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
d = "/dev/shm/emit_ab"
os.makedirs(d, exist_ok=True)
paths = []
gen = synth.haplotypes_realistic if a.realistic else synth.haplotypes_sparse
for h, bases in gen(a.haps, a.length, a.divergence, a.seed):
    p = os.path.join(d, "h%03d.fa" % h)
    synth.write_fasta_fast(p, bases, name="hap%03d" % h)
    paths.append(p)
exe = os.path.join(os.path.dirname(build.LIB), "..", "bin", "mumemto_exec")
stats = os.path.join(d, "stats.json")
names = ["text", "suffix_sort", "lcp_bwt", "scan", "verify", "rows", "windows", "total"]
order = [v for v in a.variants for _ in range(a.reps)] if a.grouped else [v for _ in range(a.reps) for v in a.variants]
for _once in (0,):
    for v in order:
        name, _, envs = v.partition("=")
        env = dict(os.environ, MUMEMTO_STATS=stats)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        if os.path.exists(stats):
            os.unlink(stats)
        time.sleep(a.pause)        # (the memory of the process before is still being wiped: a process that maps its heap too early waits for it)
        t = time.perf_counter()
        r = subprocess.run([exe, "-o", os.path.join(d, "out")] + a.args.split() + paths, capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t
        if r.returncode != 0 or not os.path.exists(stats):
            print("%-14s rc %d %s" % (name, r.returncode, r.stderr[-300:].replace("\n", " | ")), flush=True)
            continue
        if a.stderr:
            for line in r.stderr.split("\n"):
                if a.stderr in line:
                    print("    " + line, flush=True)
        st = json.load(open(stats))
        out = os.path.join(d, "out.mums" if os.path.exists(os.path.join(d, "out.mums")) else "out.mems")
        hsh = hashlib.sha256()
        with open(out, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                hsh.update(blk)
        print("%-14s %.3f s wall | %s | pfp %s | heap peak %.1f GB, mapped %.1f GB in %.3f s | sha %s rows %d" % (
            name, dt, " ".join("%s %.1f" % (n, x) for n, x in zip(names, st["stage_ms"])),
            " ".join("%.0f" % x for x in st["pfp_ms"]), st["heap_peak_bytes"] / 2**30, st["heap_mapped_bytes"] / 2**30,
            st["heap_map_seconds"], hsh.hexdigest()[:16], st["rows"]), flush=True)
shutil.rmtree(d, ignore_errors=True)
