cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r5b_kmeans
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/scripts/bench_kmeans_linear.py 64 ${1:-0} 5 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/summarize_profile.py db $DB > $O/kernel_stats.md 2>$O/sum.err
rm -rf $O/kt
tail -2 $O/kt.log; head -14 $O/kernel_stats.md
