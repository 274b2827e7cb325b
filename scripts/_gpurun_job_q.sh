cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x3_fast.py tests/test_gpu_x3.py -x -q 2>&1 | tail -4
for V in 0 1; do
timeout 300 python bench.py --steps 10 --warmup 3 --precision mixed --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $V:', d['value'], d['ms_per_step'], {k:round(v['ms_total']/10,2) for k,v in d['kernel_ms'].items()}, d.get('parity',{}).get('max_abs_tokens'))"
done
python scripts/bench_a384_x3.py 2>&1 | tail -8
