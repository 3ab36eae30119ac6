#!/bin/bash
# GPU box helper: the text-order passes of the bucket-wise producer on a mid-size packed text, kernel statistics per variant.
# usage: bash tests/micro/pass_probe.sh <tag> [length]
TAG=${1:-pass}; LEN=${2:-300000000}
R=$GRAFT_REPO_ROOT
export MUMEMTO_PRODUCER=expand MMT_PACKED_TEXT=1 MMT_GUIDED_BATCH=100000000
MMT_GUIDED_STAGE=0 bash $R/tests/profile_round6_share.sh ${TAG}_dense /root/repo/tests/big_share.py --length $LEN --no-checks > /dev/null 2>&1
MMT_GUIDED_STAGE=0 MMT_GUIDED_NO_DENSE=1 bash $R/tests/profile_round6_share.sh ${TAG}_nodense /root/repo/tests/big_share.py --length $LEN --no-checks > /dev/null 2>&1
MMT_GUIDED_STAGE=1 bash $R/tests/profile_round6_share.sh ${TAG}_staged /root/repo/tests/big_share.py --length $LEN --no-checks > /dev/null 2>&1
MMT_GUIDED_STAGE=1 MMT_GUIDED_NO_DENSE=1 bash $R/tests/profile_round6_share.sh ${TAG}_stagednodense /root/repo/tests/big_share.py --length $LEN --no-checks > /dev/null 2>&1
for v in dense nodense staged stagednodense; do echo "== $v"; grep "k_batch_fill\|k_batch_count\|k_bin_hist\|k_stage_fill\|k_stage_take\|kernel ms" $R/gpurun_out/${TAG}_$v/kernel_summary.txt; grep "^{\"rank" $R/gpurun_out/${TAG}_$v/run.log | cut -c1-200; done
