cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mlp_fused.py -x -q 2>&1 | tail -3
for o in "" "--no-fuse-proj" ""; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $o 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['step_ms']['median'], {k:round(v['ms_total']/10,2) for k,v in d['kernel_ms'].items()})"; done
