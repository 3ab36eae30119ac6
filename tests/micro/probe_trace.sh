# kernel trace of tests/realistic_probe.py: bash tests/micro/probe_trace.sh HAPS LENGTH PRODUCER
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/trp -o t --output-format csv -- python $R/tests/realistic_probe.py "$@" > /tmp/trp.log 2>&1
tail -3 /tmp/trp.log | cut -c1-400
python $R/tests/kstats.py $(find /tmp/trp -name "*kernel_stats.csv" | head -1) 2 24
