cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02n
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/summarize_profile.py db $DB > $O/kernel_stats.md 2>$O/sum.err
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-overlap > $O/pmc_$N.log 2>&1
done
python $R/scripts/summarize_profile.py pmc $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc.md 2>>$O/sum.err
rm -rf $O/kt $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -8 $O/kernel_stats.md
head -6 $O/pmc.md
cd $R
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r02n.json; cut -c1-160 gpurun_out/bench_r02n.json
timeout 300 python bench.py --precision exact 2>&1 | tail -1 > gpurun_out/bench_r02n_exact.json; cut -c1-160 gpurun_out/bench_r02n_exact.json
