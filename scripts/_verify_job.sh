cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log > gpurun_out/bench_default.json; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_default.json').read())
except Exception as e:
    print(open('gpurun_out/bench.log').read()[-3000:]); raise
print(d['value'], d['ms_per_step'], d['dtype']); print(d['config']['workload'][:300]); print(d['parity']); print(d['cpu_baseline'])
print('parity_mode', d['parity_mode']['value'], d['parity_mode']['ms_per_step'], d['parity_mode']['parity'])
print('fast', d['stego_fast']['value'], d['stego_fast']['ms_per_step'], d['stego_fast'].get('parity'))
print('roofline', d['roofline'])
PY
