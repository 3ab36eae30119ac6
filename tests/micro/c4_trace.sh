# kernel statistics of one rank's share of configs[3] ({anchor + 12} x 3.05 Gbp, merge metadata) at its size:
#   bash tests/micro/c4_trace.sh [top] [more arguments of big_share.py]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/trc4 -o t --output-format csv -- python $R/tests/big_share.py --no-checks ${2:-} > /tmp/trc4.log 2>&1
grep -E "seconds|guided\]|wgd\]" /tmp/trc4.log | tail -8
python $R/tests/kstats.py $(find /tmp/trc4 -name "*kernel_stats.csv" | head -1) 1 ${1:-40} $(find /tmp/trc4 -name "*kernel_trace.csv" | head -1)
rm -rf /tmp/trc4
