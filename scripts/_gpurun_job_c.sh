cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/check_a384_x3.py 3 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench.py --precision mixed --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline > gpurun_out/bench_mixed.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_mixed.log > gpurun_out/r04c_bench_mixed.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c_bench_mixed.json').read())
print(d['value'], d['ms_per_step']); print({k:(round(v['ms_total']/10,2), v['launches']) for k,v in d['kernel_ms'].items()})
PY
