cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -x -q 2>&1 | tail -5
PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python scripts/bench_mlp_fused.py 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['step_ms'], {k:round(v['ms_total']/8,2) for k,v in d['kernel_ms'].items()})"
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fuse-mlp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused', d['value'], d['step_ms'], {k:round(v['ms_total']/8,2) for k,v in d['kernel_ms'].items()})"
done
