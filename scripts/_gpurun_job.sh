cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gputest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log > gpurun_out/bench_default.json; cut -c1-160 gpurun_out/bench_default.json
bash scripts/profile_job.sh r03d_f16 1
bash scripts/profile_job.sh r03d_exact 0 --precision exact
bash scripts/profile_job.sh r03d_upstream 0 --stego-reading upstream
bash scripts/profile_job.sh r03d_dinov2_fp8 0 --mode dinov2
