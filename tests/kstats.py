"""Per-step summary of a rocprofv3 kernel_stats.csv: python tests/kstats.py FILE [steps] [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r['TotalDurationNs']) for r in rows)
lib = sum(int(r['TotalDurationNs']) for r in rows if 'rocprim' in r['Name'])
print("kernel ms per step %.1f   rocPRIM %.1f (%.1f %%)" % (tot / steps / 1e6, lib / steps / 1e6, 100.0 * lib / tot))
for r in rows[:top]:
    n = r['Name']
    m = re.search(r'(k_\w+|onesweep|scan_impl|segmented\w*|copyBuffer|fillBuffer|radix_sort\w*)', n)
    print("%-26s calls/step %7.1f  ms/step %8.2f  avg_us %9.1f" % (m.group(1) if m else n[:26], int(r['Calls']) / steps,
                                                                 int(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3))
