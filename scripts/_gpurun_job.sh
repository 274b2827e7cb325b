cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_attention_lazy.py -x -q 2>&1 | tail -4
for i in 1 2; do
for v in 1 3; do
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --attn-variant $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['step_ms']['median'], {k:round(v['ms_total']/8,2) for k,v in d['kernel_ms'].items()})"
done; done
