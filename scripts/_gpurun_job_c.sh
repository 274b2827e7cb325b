cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_backbone.py tests/test_gpu_attention_lazy.py -q -x -p no:cacheprovider -k "mixed or exact_mode or reference_448 or attention" -s 2>&1 | grep -v amdgpu.ids | grep "max|err|\|passed\|failed\|Error\|error" | tail -30
timeout 600 python bench.py --precision mixed --steps 10 --warmup 3 --no-extra-legs > gpurun_out/bench_mixed.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_mixed.log > gpurun_out/r04c_bench_mixed.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c_bench_mixed.json').read())
print(d['value'], d['ms_per_step'], d['dtype']); print(d['parity']); print('roofline', d['roofline']); print(d['kernel_ms'])
PY
