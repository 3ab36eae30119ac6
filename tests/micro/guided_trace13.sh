cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MUMEMTO_PRODUCER=guided rocprofv3 --kernel-trace --stats -d /tmp/trg -o t --output-format csv -- python $R/tests/big_waves.py 13 300000000 0.001 > /tmp/trg.log 2>&1
python $R/tests/kstats.py $(find /tmp/trg -name "*kernel_stats.csv" | head -1) 1 16
