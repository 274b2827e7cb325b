set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fp8.py -m gpu -q -s 2>&1 | tail -60 > gpurun_out/r02f_tests.log
python bench.py --mode dinov2 --steps 20 --warmup 5 > gpurun_out/r02f_bench_dinov2_fp8.json 2> gpurun_out/r02f_a.err
python bench.py --mode dinov2 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02f_bench_dinov2_bf16.json 2> gpurun_out/r02f_b.err
python bench.py --mode backbone --precision fp8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02f_bench_backbone_fp8.json 2> gpurun_out/r02f_c.err
cat gpurun_out/r02f_tests.log; tail -n 3 gpurun_out/r02f_*.err; cat gpurun_out/r02f_bench_*.json | cut -c1-3000
