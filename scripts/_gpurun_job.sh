set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_x3.py tests/test_gpu_supervision.py tests/test_gpu_gemm_a384.py tests/test_gpu_backbone.py tests/test_gpu_distributed.py tests/test_gpu_bridge.py tests/test_gpu_slic.py -m gpu -q -s 2>&1 | tail -120 > gpurun_out/r02c_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_fused.json 2> gpurun_out/r02c_bench_fused.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fuse-ln > gpurun_out/r02c_bench_unfused.json 2> gpurun_out/r02c_bench_unfused.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c_bench_fused2.json 2>> gpurun_out/r02c_bench_fused.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision exact > gpurun_out/r02c_bench_exact.json 2> gpurun_out/r02c_bench_exact.err
tail -30 gpurun_out/r02c_tests.log; tail -3 gpurun_out/*.err; cat gpurun_out/r02c_bench_*.json
