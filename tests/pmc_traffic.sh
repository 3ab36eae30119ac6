#!/bin/bash
# GPU box helper: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, counters only) of chosen kernels on the
# default bench.  usage: bash tests/pmc_traffic.sh '<kernel regex>' <tag>
RE=${1:-'k_lcp_gather|k_irr_lcp|k_emit'}
TAG=${2:-pmc_traffic}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$RE" -d $OUT/$c -o $c --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 > $OUT/$c.log 2>&1
done
python - $OUT <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    f = acc[k].get("FETCH_SIZE", 0) / max(1, cnt[k]["FETCH_SIZE"]); w = acc[k].get("WRITE_SIZE", 0) / max(1, cnt[k]["WRITE_SIZE"])
    res[k] = {"FETCH_SIZE_KB_per_launch_raw": f, "WRITE_SIZE_KB_per_launch_raw": w, "launches": cnt[k]["FETCH_SIZE"]}
    print("%-60s fetch %10.1f MB raw  write %10.1f MB raw  (%d launches)" % (k[:60], f / 1024, w / 1024, cnt[k]["FETCH_SIZE"]))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
