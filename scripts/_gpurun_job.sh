cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gputest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' gpurun_out/bench.log | tail -1 > gpurun_out/bench_default.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read())
print(d['value'], d['ms_per_step'], d['dtype'], d['step_ms']); print(d['parity']); print(d['cpu_baseline'])
for k in ('fp16_speed', 'parity_mode', 'stego_fast', 'backbone_b32', 'dinov2_fp8'):
    if k in d:
        print(k, d[k]['value'], d[k]['ms_per_step'], d[k]['roofline'].get('frac'), d[k].get('parity'))
print('roofline', d['roofline'])
PY
bash scripts/profile_job.sh r05d_headline 0
