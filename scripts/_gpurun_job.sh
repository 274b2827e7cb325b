cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 1500 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/suite_$i.log; grep -E "^FAILED|passed|failed" gpurun_out/suite_$i.log | cut -c1-300
done
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r02o.json; cut -c1-160 gpurun_out/bench_r02o.json
