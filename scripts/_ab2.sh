#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift; timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], {k:round(v['ms_total']/30,3) for k,v in d['kernel_ms'].items() if v['ms_total']>10})"; }
WVN_NO_HANDOVER=1 run resident
run handover
WVN_NO_HANDOVER=1 run resident
run handover
run handover_bf16 --precision bf16
timeout 400 python bench.py --no-extra-legs --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('handover+parity', d['value'], d['parity'])"
