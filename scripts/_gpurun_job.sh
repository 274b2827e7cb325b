set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap"
for PREC in bf16 exact; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02g_$PREC -o kt -- python bench.py --precision $PREC --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02g_kt_$PREC.log 2>&1
  python scripts/summarize_profile.py db $(find gpurun_out/prof_r02g_$PREC -name "*.db" | head -1) > gpurun_out/r02g_${PREC}_kernel_stats.md 2>> gpurun_out/r02g_kt_$PREC.log
  rm -rf gpurun_out/prof_r02g_$PREC
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r02g_$PREC/p1 -- $B --precision $PREC > gpurun_out/r02g_pmc1_$PREC.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r02g_$PREC/p2 -- $B --precision $PREC > gpurun_out/r02g_pmc2_$PREC.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r02g_$PREC/p3 -- $B --precision $PREC > gpurun_out/r02g_pmc3_$PREC.log 2>&1
  python scripts/summarize_profile.py pmc gpurun_out/pmc_r02g_$PREC/p1 gpurun_out/pmc_r02g_$PREC/p2 gpurun_out/pmc_r02g_$PREC/p3 > gpurun_out/r02g_${PREC}_pmc.md 2>> gpurun_out/r02g_pmc1_$PREC.log
  rm -rf gpurun_out/pmc_r02g_$PREC
done
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_quick_start.py -m gpu -q 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 > gpurun_out/r02g_bench_bf16.json 2> gpurun_out/r02g_a.err
python bench.py --precision exact --steps 20 --warmup 5 > gpurun_out/r02g_bench_exact.json 2> gpurun_out/r02g_b.err
cat gpurun_out/r02g_*_kernel_stats.md gpurun_out/r02g_*_pmc.md | cut -c1-400
