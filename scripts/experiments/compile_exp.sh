#!/bin/bash
# Compile an experiment of this directory for gfx950 (no GPU needed) and print what the register allocator made of it.
#   scripts/experiments/compile_exp.sh attention_pp -DPP_OCC=2 -DPP_XPRE=0 -DPP_LA=2
# Output: _build/exp/<name>.s (ISA), register / spill counts, AGPR traffic, scratch use.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/_build/exp" && cd "$ROOT/_build/exp"
cp "$ROOT/scripts/experiments/$NAME.hip.txt" "$NAME.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form \
  -I"$ROOT/wild_visual_navigation_amd/csrc" -I"$ROOT/include" "$@" -S --cuda-device-only "$NAME.hip" -o "$NAME.s"
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_count|private_segment_fixed_size):" "$NAME.s" | head -4
echo "v_accvgpr: $(grep -c v_accvgpr "$NAME.s")  scratch: $(grep -c scratch_ "$NAME.s")  mfma: $(grep -c v_mfma "$NAME.s")  lines: $(wc -l < "$NAME.s")"
