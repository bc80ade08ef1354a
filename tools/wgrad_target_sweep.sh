#!/bin/bash
# Workgroups per weight-gradient launch (wgrad_plan's target; fewer, longer splits = less partial-sum traffic and a shorter
# wgrad_reduce4, more = better fill of the last round): PCS_WGRAD_TARGET sweep over tools/wgrad_microbench.py (kernel + reduce).
cd "$(dirname "$0")/.."
for t in 0 768 1024 1536 2048 3072 4096 6144; do
  echo "== PCS_WGRAD_TARGET=$t (0 = the library's plan)"
  PCS_WGRAD_TARGET=$t python tools/wgrad_microbench.py 2>/dev/null | grep -v "^|--" | awk -F'|' 'NR>2 && NF>8 {printf "%s %s  bf16 %s ms  fp32 %s ms\n", $2, $5, $9, $6} /total/ {print}'
done
