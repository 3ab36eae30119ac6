# kernel trace + PMC passes (FETCH_SIZE, WRITE_SIZE: counters only, separate runs) of k_scan in the configs[4] instantiation
# (-k -1 -f 3, window 92, walks up to 282) on the C3 stand-in:  bash tests/micro/c5_scan_profile.sh <tag>  -> gpurun_out/<tag>/
TAG=${1:-round5_c5scan}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $R/tests/mem_mode_c3.py > $OUT/run.log 2>&1
grep '"mode"' $OUT/run.log | tail -2
python $R/tests/kstats.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 2 12 $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_summary.txt
grep -E "k_scan|busy" $OUT/kernel_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex 'mmt::k::k_scan' -d $OUT/pmc_$c -o $c --output-format csv -- \
      python $R/tests/mem_mode_c3.py 94 64000000 1 > $OUT/pmc_$c.log 2>&1
  cp $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) $OUT/scan_${c}_counter_collection.csv
done
python - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
n = 12032000188
def total(name):
    rows = [r for r in csv.DictReader(open("%s/scan_%s_counter_collection.csv" % (out, name))) if r["Counter_Name"] == name]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows), rows[0]["Kernel_Name"].split("(")[0]
f, nf, kn = total("FETCH_SIZE"); w, nw, _ = total("WRITE_SIZE")
run = [json.loads(l) for l in open(out + "/run.log") if l.startswith("{")][-1]
d = {"workload": "94 haplotypes x 64,000,000 bp (C3 stand-in), -k -1 -f 3: the scan instantiation of BASELINE configs[4]",
     "kernel": kn, "launches": nf, "FETCH_SIZE_kb_raw": f, "WRITE_SIZE_kb": w,
     "hbm_bytes_per_pass": (2.0 * f + w) * 1024.0, "correction": "FETCH_SIZE x2 (gfx950: 16-byte-per-lane streaming reads are under-reported by 2x)",
     "algorithmic_bytes_per_pass": 10 * n, "scan_kernel_ms": run["scan_kernel_ms"], "roofline_frac_by_the_formula": run["roofline_frac"],
     "frac_moved": (2.0 * f + w) * 1024.0 / (run["scan_kernel_ms"] * 1e-3) / 8e12}
json.dump(d, open(out + "/scan_pmc.json", "w"), indent=1)
print(json.dumps(d))
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
