cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x3_fast.py tests/test_gpu_x3.py -x -q 2>&1 | tail -3
for V in 0 0; do
WVN_X3_DEBUG_BITS=$V timeout 300 python bench.py --steps 10 --warmup 3 --precision mixed --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bits=$V:', d['value'], d['ms_per_step'], {k:round(v['ms_total']/10,2) for k,v in d['kernel_ms'].items()}, d['parity']['max_abs_tokens'] if 'parity' in d else '')"
done
timeout 300 python bench.py --steps 6 --warmup 2 --precision exact --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact:', d['value'], d['ms_per_step'], {k:round(v['ms_total']/6,2) for k,v in d['kernel_ms'].items()})"
