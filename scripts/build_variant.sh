#!/bin/bash
# Build a variant of the library for same-box A/B runs (scripts/ab_lib.sh, WVN_LIB_PATH):
#   scripts/build_variant.sh <name> <source.hip> <extra hipcc flags...>   ->  wild_visual_navigation_amd/lib/libwvn_<name>.so
# The named source is recompiled (both operand formats when it is a dual-operand source) with the extra flags; every other object is
# taken from csrc/_build as the last regular build left it.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
C=wild_visual_navigation_amd/csrc; B=$C/_build; V=$B/variant_$name; mkdir -p $V
base=${src%.hip}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wall -Wno-unused-function"
case $src in attention_bf16.hip|attention_x3.hip) FL="$FL -fno-honor-nans";; stego.hip|stego_linear.hip|supervision.hip) FL="$FL -ffp-contract=off";; esac
/opt/rocm/bin/hipcc $FL "$@" -c $C/$src -o $V/$base.o &
if [ -f $B/${base}_f16.o ]; then /opt/rocm/bin/hipcc $FL "$@" -DWVN_OPERAND_F16=1 -c $C/$src -o $V/${base}_f16.o & fi
wait
objs=""
for o in $B/*.o; do b=$(basename $o); if [ -f $V/$b ]; then objs="$objs $V/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o wild_visual_navigation_amd/lib/libwvn_$name.so $objs
echo "built wild_visual_navigation_amd/lib/libwvn_$name.so"
