#!/bin/bash
# Same-box A/B of library builds / kernel variants (one gpurun call): every line is one bench.py headline leg.
#   scripts/ab_lib.sh "<label> <lib file under wild_visual_navigation_amd/lib> <extra bench args>" ...
cd ${GRAFT_REPO_ROOT:-.}
for spec in "$@"; do
  set -- $spec; label=$1; libf=$2; shift 2
  WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/$libf timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], {k:round(v['ms_total']/30,3) for k,v in d['kernel_ms'].items() if v['ms_total']>10})"
done
