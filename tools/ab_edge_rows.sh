#!/bin/bash
# strip heights of the 4-pixels-per-lane edge kernel (variants e4 / e5 / e6 / e8 = -DSM_EDGE4_ROWS=k), edges alone and the whole step
for cfg in "C3 1" "C5 1" "C4 8" "C2 1"; do
  set -- $cfg
  AB_EDGES=1 timeout -k 10 200 python tools/ab_variants.py $1 $2 9
  AB_STEP=1 timeout -k 10 200 python tools/ab_variants.py $1 $2 9 | sed 's/^/step: /'
done 2>&1 | grep -v amdgpu.ids
