#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() { label=$1; shift; timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['step_ms'])"; }
run headline; run headline
