#!/bin/bash
# GPU box helper: rocprofv3 kernel trace of the default bench, top kernels by time per step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_top; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample-bp 0 ${EXTRA} > $OUT/log 2>&1
python - $OUT/t_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms per step %.1f" % (tot / 4 / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 28]:
    print("%7.2f ms %5.1f%% x%-5.1f %s" % (float(r['TotalDurationNs']) / 4e6, 100 * float(r['TotalDurationNs']) / tot, int(r['Calls']) / 4, r['Name'][:100]))
PY
