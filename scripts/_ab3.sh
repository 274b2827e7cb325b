#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; lib=$2; shift 2; WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/libwvn_$lib.so timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], {k:round(v['ms_total']/30,3) for k,v in d['kernel_ms'].items() if v['ms_total']>10})"; }
for i in 1 2; do for l in "$@"; do run $l $l; done; done
