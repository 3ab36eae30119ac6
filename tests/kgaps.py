"""Idle time of the device between kernels in a rocprofv3 kernel_trace.csv, attributed to the kernel BEFORE the gap:
python tests/kgaps.py FILE [steps] [top]   (steps = passes over the workload in the trace, for per-step figures)"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)


def short(n):
    return n.split("(")[0].replace("void ", "").replace("mmt::", "").split("<")[0][-44:]


gap_by, cnt_by = defaultdict(float), defaultdict(int)
busy_end, busy, idle = ev[0][1], 0.0, 0.0
last = ev[0]
busy += ev[0][1] - ev[0][0]
for s, e, n in ev[1:]:
    if s > busy_end:
        g = s - busy_end
        if g < 50e6:                       # (gaps of more than 50 ms: between the steps / legs of the bench)
            idle += g
            gap_by[short(last[2]) + " -> " + short(n)] += g
            cnt_by[short(last[2]) + " -> " + short(n)] += 1
        busy += e - s
    elif e > busy_end:
        busy += e - busy_end
    if e > busy_end:
        busy_end, last = e, (s, e, n)
print("device busy %.1f ms per step, idle between kernels %.1f ms per step (gaps below 50 ms)" % (busy / steps / 1e6, idle / steps / 1e6))
for k, v in sorted(gap_by.items(), key=lambda kv: -kv[1])[:top]:
    print("%8.2f ms per step  x%-6.1f %s" % (v / steps / 1e6, cnt_by[k] / steps, k))
