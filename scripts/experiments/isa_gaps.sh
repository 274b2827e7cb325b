#!/bin/bash
# Print the first loop of an ISA listing as one line per MFMA "gap": what is issued between two consecutive MFMAs, in order.
#   scripts/experiments/isa_gaps.sh _build/exp/attention_pp.s [lines]
S=$1; L=$(grep -n "Loop Header" "$S" | head -1 | cut -d: -f1)
awk -v a="$L" 'NR>=a && NR<=a+450' "$S" | grep -v "^\s*;" | awk '{ if ($1 ~ /v_mfma/) {printf "\nMFMA | "} else if ($1 ~ /s_waitcnt/) {printf "W[%s] ", $2} else printf "%s ", $1 }' |
  sed 's/v_exp_f32_e32/EXP/g; s/v_cvt_pk_bf16_f32/CVT/g; s/v_dot2c_f32_bf16_e32/DOT/g; s/ds_read_b128/DS/g; s/v_add_u32_e32/ADD/g; s/buffer_load_dwordx4/DMA/g; s/v_accvgpr_write_b32/AW/g; s/v_accvgpr_read_b32/AR/g; s/s_nop/nop/g' | head -"${2:-40}"
echo
