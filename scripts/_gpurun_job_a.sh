cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gputest.log | tail -12
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log > gpurun_out/r04d_bench_default.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04d_bench_default.json').read())
print(d['value'], d['ms_per_step'], d['dtype'], d['step_ms']); print(d['parity']); print(d['cpu_baseline'])
print('parity_mode', d['parity_mode']['value'], d['parity_mode']['ms_per_step'], d['parity_mode']['parity'])
print('fast', d['stego_fast']['value'], d['stego_fast']['ms_per_step'], d['stego_fast'].get('parity'))
print('roofline', d['roofline']); print(d['kernel_ms'])
PY
