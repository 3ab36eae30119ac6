cd /tmp && hipcc -O2 --offload-arch=gfx950 $GRAFT_REPO_ROOT/tests/micro/exit_probe.cpp -o /tmp/exit_probe 2>/dev/null
for args in "0 0 0" "0 0 0" "100 0 0" "0 6 0" "0 0 32"; do
  sleep 6
  python3 - "$args" <<'PY'
import subprocess, sys, time
a = sys.argv[1].split()
t = time.perf_counter(); r = subprocess.run(["/tmp/exit_probe"] + a, capture_output=True, text=True); dt = time.perf_counter() - t
print("map %3s GB, pinned %s GB, %2s threads: %.3f s wall (%s)" % (a[0], a[1], a[2], dt, r.stderr.strip()))
PY
done
