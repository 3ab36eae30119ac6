"""GPU box helper: wall-clock of the command line on the bench workload (FASTA in page cache -> PREFIX.mums)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mumemto_amd import synth, build
haps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12_100_000
d = "/tmp/cli_timing"; os.makedirs(d, exist_ok=True)
paths = []
if len(sys.argv) > 3 and sys.argv[3] == "bench":            # the collection of the bench line (bench.py defaults)
    for h, bases in synth.haplotypes_sparse(haps, L, 0.001, 3):
        p = os.path.join(d, "h%02d.fa" % h); synth.write_fasta_fast(p, bases, name="hap%03d" % h); paths.append(p)
else:
    for i, doc in enumerate(synth.pangenome(haps, L, 0.005, 2)):
        p = os.path.join(d, "h%02d.fa" % i); synth.write_fasta(p, doc, width=80); paths.append(p)
exe = os.path.join(os.path.dirname(build.LIB), "..", "bin", "mumemto_exec")
reps = int(os.environ.get("CLI_REPS", "3"))
for rep in range(reps):
    extra = {}
    t = time.perf_counter()
    r = subprocess.run([exe, "-o", os.path.join(d, "out")] + paths, capture_output=True, text=True,
                       env=dict(os.environ, MUMEMTO_TIMING="1", **extra))
    dt = time.perf_counter() - t
    print("run %d: %.3f s wall, rc %d, %.3f Gbp/s" % (rep, dt, r.returncode, haps * L / dt / 1e9))
    print("\n".join(l for l in r.stderr.split("\n") if "sec" in l or "stages" in l or "[timing]" in l or "[mem]" in l))
print(os.path.getsize(os.path.join(d, "out.mums")), "bytes of .mums")
t = time.perf_counter(); subprocess.run([exe], capture_output=True); print("no arguments (load + exit): %.3f s" % (time.perf_counter() - t))
t = time.perf_counter()
subprocess.run([exe, "-o", os.path.join(d, "out")] + paths, capture_output=True, env=dict(os.environ, MUMEMTO_FULL_TEARDOWN="1"))
print("with the orderly teardown: %.3f s wall" % (time.perf_counter() - t))
