#!/bin/bash
# Same-box A/B of library builds on the fused block kernels alone (one gpurun call): scripts/ab_mlp_fused.sh hip <variant> ...
# (variants: scripts/build_variant.sh <name> mlp_fused.hip -D...; "hip" = the regular build)
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
for l in "$@"; do WVN_LIB_PATH=$PWD/wild_visual_navigation_amd/lib/libwvn_$l.so timeout 120 python scripts/mlp_fused_ab.py $l 2>&1 | grep -v amdgpu.ids | tail -2; done
