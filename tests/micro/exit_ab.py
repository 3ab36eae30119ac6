"""GPU box helper: process start -> exit of mumemto_exec on the bench workload, with and without handing the free top of the
device heap back while the run goes on (MUMEMTO_NO_EARLY_UNMAP=1).  usage: exit_ab.py [reps]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mumemto_amd import synth, build
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = "/dev/shm/exit_ab"; os.makedirs(d, exist_ok=True)
paths = []
for h, bases in synth.haplotypes_sparse(94, 64_000_000, 0.001, 3):
    p = os.path.join(d, "h%02d.fa" % h); synth.write_fasta_fast(p, bases, name="hap%03d" % h); paths.append(p)
exe = os.path.join(os.path.dirname(build.LIB), "..", "bin", "mumemto_exec")
for rep in range(reps):
    for mode, env in (("upload overlapped", {"MUMEMTO_NO_EARLY_UNMAP": "1"}), ("upload after the reads", {"MUMEMTO_NO_EARLY_UNMAP": "1", "MUMEMTO_NO_UPLOAD_OVERLAP": "1"})):
        time.sleep(8)
        t = time.perf_counter()
        r = subprocess.run([exe, "-o", os.path.join(d, "out")] + paths, capture_output=True, text=True, env=dict(os.environ, MUMEMTO_TIMING="1", **env))
        dt = time.perf_counter() - t
        marks = [l for l in r.stderr.split("\n") if "[timing]" in l]
        print("%-18s %.3f s wall, rc %d | %s" % (mode, dt, r.returncode, " | ".join(m.replace("[timing]", "").strip() for m in marks[-3:])), flush=True)
import shutil; shutil.rmtree(d, ignore_errors=True)
