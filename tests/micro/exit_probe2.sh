# GPU box: exit latency (wall clock of the parent minus the child's own clock at _Exit) by kind of memory held
# usage: exit_probe2.sh ["mode GB" ...]
cd /tmp && hipcc -O2 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tests/micro/exit_probe2.cpp -o /tmp/exit_probe2 2>/dev/null
if [ $# -eq 0 ]; then set -- "none 0" "none 0" "malloc 120" "vmm 120" "vmm 120" "vmm 60" "vmm_unmap 120" "host 6" "malloc 120"; fi
for args in "$@"; do
  sleep 7
  python3 - "$args" <<'PY'
import subprocess, sys, time
a = sys.argv[1].split()
t = time.perf_counter(); r = subprocess.run(["/tmp/exit_probe2"] + a, capture_output=True, text=True); dt = time.perf_counter() - t
err = r.stderr.strip()
up = float(err.split()[-1]) if r.returncode == 0 and err else -1
print("%-14s %4s GB: %.3f s wall, %.3f s to _Exit, %.3f s after it (rc %d) %s" % (a[0], a[1], dt, up, dt - up, r.returncode, err.rsplit(";", 1)[0] if ";" in err else ""), flush=True)
PY
done
