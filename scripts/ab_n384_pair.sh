cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_x3_fast.py tests/test_gpu_x3.py tests/test_gpu_robustness.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for pair in 1 0 1 0; do WVN_N384_PAIR=$pair timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pair=$pair', d['value'], d['ms_per_step'], {k:round(v['ms_total']/20,2) for k,v in d['kernel_ms'].items() if v['ms_total']>100})"; done
