cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/debug_pixel_kmeans3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/debug_pix3.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/gputest.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-300
