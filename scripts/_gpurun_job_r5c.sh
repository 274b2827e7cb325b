cd $GRAFT_REPO_ROOT
python scripts/bench_kmeans_linear.py 64 4 5
for v in 1 2 3 4 7 8 15; do echo "ABL=$v"; WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/libwvn_abl$v.so python scripts/bench_kmeans_linear.py 64 4 5 2>/dev/null; done
