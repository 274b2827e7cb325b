cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stego_linear.py -q --tb=short -p no:cacheprovider -x > gpurun_out/r5a_linear_tests.log 2>&1; echo "linear tests rc=$?"; tail -15 gpurun_out/r5a_linear_tests.log
timeout 300 python scripts/bench_pixel_kmeans.py 64 > gpurun_out/r5a_kmeans_bench.log 2>&1; echo "kmeans bench rc=$?"; cat gpurun_out/r5a_kmeans_bench.log
