#!/bin/bash
# GPU box helper: default bench line (C3 stand-in) + rocprofv3 kernel trace of the same command (statistics + union times,
# tests/kstats.py) + the PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, separate runs) for k_scan AND for k_emit,
# summarised into scan_pmc.json / emit_pmc.json.
# usage: bash tests/profile_round6.sh <tag> [bench|trace|pmc|all]   -> gpurun_out/<tag>/
TAG=${1:-round6_x}
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
      python $R/bench.py --steps 2 --warmup 1 --no-extras --in-process > $OUT/trace.log 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
  python $R/tests/kstats.py $OUT/kernel_stats.csv 3 40 $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_summary.txt
  head -24 $OUT/kernel_summary.txt
  python $R/tests/kgaps.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 3 8 > $OUT/device_idle_gaps.txt
  head -3 $OUT/device_idle_gaps.txt
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  for k in scan emit; do
    re='mmt::k::k_scan'; [ $k = emit ] && re='mmt::pk::k_emit2<'
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$re" -d $OUT/pmc_${k}_$c -o $c --output-format csv -- \
          python $R/bench.py --steps 1 --warmup 1 --no-extras --in-process > $OUT/pmc_${k}_$c.log 2>&1
      cp $(find $OUT/pmc_${k}_$c -name "*counter_collection.csv" | head -1) $OUT/${k}_${c}_counter_collection.csv
    done
  done
  python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
n = 12032000188
def total(kern, name):
    rows = list(csv.DictReader(open("%s/%s_%s_counter_collection.csv" % (out, kern, name))))
    rows = [r for r in rows if r["Counter_Name"] == name]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows), rows[0]["Kernel_Name"]
for kern, algo, what in (("scan", 10 * n, "SA 5 + LCP 4 + BWT 1 bytes per suffix read"),
                         ("emit", 10 * n + 12 * n, "10 bytes per suffix written + 12 bytes of occurrence record per suffix read")):
    f, nf, kn = total(kern, "FETCH_SIZE")
    w, nw, _ = total(kern, "WRITE_SIZE")
    steps = 2.0                       # --steps 1 --warmup 1: two passes over the stream
    d = {"workload": "bench.py default: 94 haplotypes x 64,000,000 bp, divergence 0.001, seed 3 (|T| = 12,032,000,188)",
         "kernel": kn.split("(")[0], "launches_counted": nf, "launches_per_step": nf / steps,
         "FETCH_SIZE_kb_raw_per_step": f / steps, "WRITE_SIZE_kb_per_step": w / steps,
         "correction": "FETCH_SIZE x2 for k_scan (gfx950 note of MI355X_MICROARCH.md: wide coalesced streaming reads, 16 B per lane, are under-reported by 2x); raw for k_emit (8 / 4 B per lane gathers: uncalibrated width, the x2 figure beside it as the upper bound); WRITE_SIZE as reported; KB = 1024 B",
         # k_scan streams 16 bytes per lane: the guide's x2 applies.  k_emit gathers 8 and 4 bytes per lane in runs of ~750
         # bytes: "other access widths are uncalibrated" -- its figure is the raw one, the x2 form is the upper bound
         "hbm_bytes_per_step": ((2.0 if kern == "scan" else 1.0) * f + w) / steps * 1024.0,
         "hbm_bytes_per_step_fetch_x2": (2.0 * f + w) / steps * 1024.0, "hbm_bytes_per_step_fetch_raw": (f + w) / steps * 1024.0,
         "algorithmic_bytes_per_step": algo, "algorithmic_bytes": what,
         "recipe": "tests/profile_round6.sh (two separate rocprofv3 --kernel-trace --pmc passes per kernel, counters only)"}
    json.dump(d, open("%s/%s_pmc.json" % (out, kern), "w"), indent=1)
    print(json.dumps(d))
PY
fi
rm -rf $OUT/trace $OUT/pmc_scan_FETCH_SIZE $OUT/pmc_scan_WRITE_SIZE $OUT/pmc_emit_FETCH_SIZE $OUT/pmc_emit_WRITE_SIZE
