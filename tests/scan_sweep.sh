#!/bin/bash
# scan-kernel variant sweep (GPU box): prints avg kernel ms per variant
# usage: scan_sweep.sh <haps> <length> [variants] [blocks-per-CU list]
for v in ${3:-0 1 2 3 4 5 6}; do for g in ${4:-8}; do
  r=$(MMT_SCAN_VARIANT=$v MMT_SCAN_BPC=$g python bench.py --steps 3 --warmup 1 --cpu-sample-bp 0 --haps ${1:-16} --length ${2:-12100000} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], round(d['roofline']['frac'],4), d['config']['output_rows'])")
  echo "variant=$v bpc=$g -> $r"
done; done
