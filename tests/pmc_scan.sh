#!/bin/bash
# GPU box helper: SQ counter passes for one kernel (separate rocprofv3 runs, counters only).
# usage: pmc_scan.sh [kernel-name-substring, default mmt::k::k_scan]
export KERNEL=${1:-mmt::k::k_scan}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
run() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- \
     python $R/bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections, os
KERNEL = os.environ['KERNEL']
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if KERNEL not in k: continue
    acc[KERNEL][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[r["Counter_Name"]] += 1
for c, v in acc[KERNEL].items():
    print("%-28s %16.0f per launch (%d launches)" % (c, v / cnt[c], cnt[c]))
PY
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
