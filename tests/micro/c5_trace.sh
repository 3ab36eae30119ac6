# kernel statistics of a rank's share of a packed-text run with the modulus a 250 G-character text gets:
#   bash tests/micro/c5_trace.sh [packed 0|1] [p (0: automatic)] [haps] [length] [top] [more arguments of big_c5.py]
# (MMT_GUIDED_NO_RANK=1 in the environment: element records without the parse rank, as beyond ~2^33 characters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MMT_PACKED_TEXT=${1:-1}
P=${2:-157}
H=${3:-41}
L=${4:-100000000}
WP="--wp 14 $P"; [ "$P" = 0 ] && WP=""
rocprofv3 --kernel-trace --stats -d /tmp/trc -o t --output-format csv -- python $R/tests/big_c5.py --haps $H --length $L $WP --samples 20 ${6:-} > /tmp/trc.log 2>&1
grep -E "seconds|guided\]" /tmp/trc.log | tail -5
python $R/tests/kstats.py $(find /tmp/trc -name "*kernel_stats.csv" | head -1) 1 ${5:-14} $(find /tmp/trc -name "*kernel_trace.csv" | head -1)
rm -rf /tmp/trc
