cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r04a_kmeans
rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/bench_pixel_kmeans.py > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/summarize_profile.py db $DB > $O/kernel_stats.md 2>$O/sum.err
tail -4 $O/kt.log | head -3
head -9 $O/kernel_stats.md
rm -rf $O/kt
