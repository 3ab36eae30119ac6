#!/bin/bash
for c in 16 32 64 128 256; do for b in 64 256; do
  r=$(MMT_LCP_CHUNK=$c MMT_LCP_BLOCK=$b python bench.py --steps 2 --warmup 1 --cpu-sample-bp 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['stage_ms_avg']['lcp_bwt'],2), round(d['pfp']['last_step_ms']['dictionary_lcp_groups'],2), round(d['ms_per_step'],1))")
  echo "chunk=$c block=$b -> $r"
done; done
