#!/bin/bash
# GPU box helper: default bench line (C3 stand-in) + rocprofv3 kernel trace of the same command + the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; counters only, separate runs) for k_scan, summarised into scan_pmc.json.
# usage: bash tests/profile_round3.sh <tag> [bench|trace|pmc|all]   -> gpurun_out/<tag>/
TAG=${1:-round3_x}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  timeout 1200 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
  tail -c 300 $OUT/bench.json
fi
if [ "$WHAT" = all ] || [ "$WHAT" = trace ]; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --no-extras > $OUT/trace.log 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
  python $R/tests/kstats.py $OUT/kernel_stats.csv 3 12
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan' -d $OUT/pmc_$c -o $c --output-format csv -- \
        python $R/bench.py --steps 1 --warmup 1 --no-extras > $OUT/pmc_$c.log 2>&1
    cp $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv
  done
  python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
def total(name):
    rows = list(csv.DictReader(open("%s/%s_counter_collection.csv" % (out, name))))
    rows = [r for r in rows if r["Counter_Name"] == name]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows), rows[0]["Kernel_Name"]
f, nf, kn = total("FETCH_SIZE")
w, nw, _ = total("WRITE_SIZE")
steps = 2.0                       # --steps 1 --warmup 1: two passes over the stream
n = 12032000188
d = {"workload": "bench.py default: 94 haplotypes x 64,000,000 bp, divergence 0.001, seed 3 (|T| = 12,032,000,188)",
     "kernel": kn.split("(")[0], "launches_counted": nf, "launches_per_step": nf / steps,
     "FETCH_SIZE_kb_raw_per_step": f / steps, "WRITE_SIZE_kb_per_step": w / steps,
     "correction": "FETCH_SIZE x2 (gfx950 note of MI355X_MICROARCH.md: wide coalesced reads are under-reported by 2x), WRITE_SIZE as reported; KB = 1024 B",
     "hbm_bytes_per_step": (2.0 * f + w) / steps * 1024.0, "algorithmic_bytes_per_step": 10 * n,
     "recipe": "tests/profile_round3.sh (two separate rocprofv3 --kernel-trace --pmc passes, counters only)"}
json.dump(d, open(out + "/scan_pmc.json", "w"), indent=1)
print(json.dumps(d))
PY
fi
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
