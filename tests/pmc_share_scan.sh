#!/bin/bash
# GPU box helper: PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, separate runs) of k_scan and k_verify_packed on ONE rank's share of
# configs[3] (merge-metadata instantiation of the scan: 13 documents, EXACT + ALL).   usage: bash tests/pmc_share_scan.sh <tag>
TAG=${1:-share_scan_pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan|k_verify_packed' -d $OUT/pmc_$c -o $c --output-format csv -- \
      python $R/tests/big_share.py --no-checks > $OUT/pmc_$c.log 2>&1
  cp $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv
  cp $(find $OUT/pmc_$c -name "*kernel_trace.csv" | head -1) $OUT/${c}_kernel_trace.csv 2>/dev/null
  rm -rf $OUT/pmc_$c
done
python - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (out, name))):
        if r["Counter_Name"] != name: continue
        k = "k_scan" if "k_scan" in r["Kernel_Name"] else "k_verify_packed"
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in tot: res.setdefault(k, {})[name + "_kb"] = tot[k]; res[k]["launches"] = cnt[k]
dur = collections.defaultdict(float)
try:
    for r in csv.DictReader(open("%s/FETCH_SIZE_kernel_trace.csv" % out)):
        k = "k_scan" if "k_scan" in r["Kernel_Name"] else ("k_verify_packed" if "k_verify_packed" in r["Kernel_Name"] else None)
        if k: dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
except Exception as e:
    res["trace_error"] = str(e)
n = 79300000026
for k in res:
    if not isinstance(res[k], dict): continue
    f, w = res[k].get("FETCH_SIZE_kb", 0), res[k].get("WRITE_SIZE_kb", 0)
    res[k]["hbm_bytes_fetch_x2"] = (2 * f + w) * 1024; res[k]["hbm_bytes_fetch_raw"] = (f + w) * 1024
    res[k]["ms_under_the_counters"] = dur.get(k)
res["algorithmic_bytes_k_scan"] = 10 * n
res["workload"] = "tests/big_share.py: rank 0's share of configs[3], {anchor + 12} x 3.05 Gbp, 79,300,000,026 suffixes, merge metadata"
json.dump(res, open("%s/share_scan_pmc.json" % out, "w"), indent=1)
print(json.dumps(res))
PY
