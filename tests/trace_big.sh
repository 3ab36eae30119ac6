#!/bin/bash
# GPU box helper: kernel trace of the 94 x 20 Mbp workload, top kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_big; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/bench.py --haps 94 --length 20000000 --divergence 0.001 --seed 3 --steps 2 --warmup 1 --cpu-sample-bp 0 > $OUT/log 2>&1
python - $OUT/t_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms per step %.1f" % (tot / 3 / 1e6))
for r in rows[:30]:
    print("%7.2f ms %5.1f%% x%-5.1f %s" % (float(r['TotalDurationNs']) / 3e6, 100 * float(r['TotalDurationNs']) / tot, int(r['Calls']) / 3, r['Name'][:100]))
PY
