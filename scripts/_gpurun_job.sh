cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r02m.json; cut -c1-200 gpurun_out/bench_r02m.json
