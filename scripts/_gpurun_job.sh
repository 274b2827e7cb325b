set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench_bf16.json 2> gpurun_out/r02a_bench_bf16.err
python bench.py --precision exact --steps 20 --warmup 5 > gpurun_out/r02a_bench_exact.json 2> gpurun_out/r02a_bench_exact.err
python bench.py --mode backbone --steps 20 --warmup 5 > gpurun_out/r02a_bench_backbone.json 2> gpurun_out/r02a_bench_backbone.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02a_exact -o r02a -- python bench.py --precision exact --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_prof_exact.log 2>&1
python scripts/summarize_profile.py db $(find gpurun_out/prof_r02a_exact -name "*.db" | head -1) > gpurun_out/r02a_exact_kernel_stats.md 2>> gpurun_out/r02a_prof_exact.log
rm -rf gpurun_out/prof_r02a_exact
tail -3 gpurun_out/*.err; cat gpurun_out/r02a_bench_*.json; cat gpurun_out/r02a_exact_kernel_stats.md
