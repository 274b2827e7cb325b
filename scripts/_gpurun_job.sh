set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r02e_tests.log
python bench.py --mode dinov2 --steps 20 --warmup 5 > gpurun_out/r02e_bench_dinov2_fp8.json 2> gpurun_out/r02e_a.err
python bench.py --mode dinov2 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02e_bench_dinov2_bf16.json 2> gpurun_out/r02e_b.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02e_bench_bf16.json 2> gpurun_out/r02e_c.err
tail -15 gpurun_out/r02e_tests.log; tail -n 3 gpurun_out/r02e_*.err; cat gpurun_out/r02e_bench_*.json | cut -c1-1800
