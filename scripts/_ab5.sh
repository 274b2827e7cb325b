#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
for l in "$@"; do echo -n "$l: "; WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/libwvn_$l.so python scripts/bench_pixel_kmeans.py | tail -1; done
