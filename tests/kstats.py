"""Per-step summary of a rocprofv3 run: python tests/kstats.py KERNEL_STATS_CSV [steps] [top] [KERNEL_TRACE_CSV]

Without a trace: per kernel the calls, the SUM of the durations of its launches and their average -- sums count two kernels
that run side by side on two streams twice.  With the kernel trace (the fourth argument) every kernel also gets its
WALL-COVERED time: the wall time during which at least one launch of it ran, and its share of the device's busy time when
overlap is split evenly between the kernels that run at the same moment -- the shares add up to the time the device was
busy, so a kernel that only looks long because it runs beside another one (k_flag_unsorted beside k_scatter_rank on the
sorter's second stream) stops inflating the total."""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
trace = sys.argv[4] if len(sys.argv) > 4 else None


def short(n):
    m = re.search(r'(k_\w+|onesweep|scan_impl|segmented\w*|copyBuffer|fillBuffer|radix_sort\w*)', n)
    return m.group(1) if m else n[:26]


tot = sum(int(r['TotalDurationNs']) for r in rows)
lib = sum(int(r['TotalDurationNs']) for r in rows if 'rocprim' in r['Name'])
print("kernel ms per step %.1f (sum of launch durations)   rocPRIM %.1f (%.1f %%)" % (tot / steps / 1e6, lib / steps / 1e6, 100.0 * lib / tot))
share, covered = {}, {}
if trace:
    ev = []
    for r in csv.DictReader(open(trace)):
        n = short(r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), 1, n)); ev.append((int(r["End_Timestamp"]), -1, n))
    ev.sort(key=lambda e: (e[0], e[1]))
    running = defaultdict(int)
    share, covered = defaultdict(float), defaultdict(float)
    busy, prev = 0.0, None
    for t, d, n in ev:
        if prev is not None and t > prev:
            live = [k for k, c in running.items() if c > 0]
            if live:
                busy += t - prev
                for k in live:
                    covered[k] += t - prev
                    share[k] += (t - prev) / len(live)
        running[n] += d
        prev = t
    lib_share = sum(v for k, v in share.items() if k in ("onesweep", "scan_impl") or k.startswith(("segmented", "radix_sort")))
    print("device busy %.1f ms per step (union of all launches); rocPRIM share of it %.1f ms (%.1f %%)"
          % (busy / steps / 1e6, lib_share / steps / 1e6, 100.0 * lib_share / max(busy, 1)))
agg = defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[short(r['Name'])]
    a[0] += int(r['Calls']); a[1] += int(r['TotalDurationNs'])
for n, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    line = "%-26s calls/step %7.1f  sum ms/step %8.2f  avg_us %9.1f" % (n, calls / steps, ns / steps / 1e6, ns / max(calls, 1) / 1e3)
    if trace:
        line += "  wall-covered ms/step %8.2f  share of busy ms/step %8.2f" % (covered.get(n, 0) / steps / 1e6, share.get(n, 0) / steps / 1e6)
    print(line)
