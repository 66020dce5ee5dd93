#!/bin/bash
# re-phasing s_nop levels of the bit-sliced kernel (variants nop3 / nop2 / nop1 = -DSM_BS_NOP=k), real step
export AB_STEP=1
for c in "C3 1" "C5 1" "C4 8" "C2 1" "REF4K 1"; do
  set -- $c
  timeout -k 10 200 python tools/ab_variants.py $1 $2 9 || exit 1
done 2>&1 | grep -v amdgpu.ids
