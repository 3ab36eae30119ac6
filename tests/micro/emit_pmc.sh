# SQ counters of the emitter on the default bench workload (one pass): where do its wave cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-k_emit}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-include-regex "$K" -d /tmp/pmc -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-extras > /tmp/pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in acc: print("%-22s %16.0f  (%d records)" % (k, acc[k], n[k]))
PY
