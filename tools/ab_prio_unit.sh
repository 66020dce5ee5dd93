#!/bin/bash
# the priority schedule's unit (log2 cycles) and pattern on short launches, match loop and real step
export AB_ENVS="SM_PATTERN=0xAAAAAAAA;SM_PATTERN=0xAAAAAAAA,SM_PRIO_UNIT=13;SM_PATTERN=0xAAAAAAAA,SM_PRIO_UNIT=12;SM_PATTERN=0xAAAAAAAA,SM_PRIO_UNIT=11;SM_PATTERN=0xCCCCCCCC,SM_PRIO_UNIT=12;SM_PRIO_UNIT=12;SM_PRIO_UNIT=13"
for step in "" 1; do
  export AB_STEP=$step
  echo "== AB_STEP=${step:-0}"
  for c in "C2 1" "C1 1" "REF4K 1" "C3 1"; do
    set -- $c
    timeout -k 10 200 python tools/ab_variants.py $1 $2 9 || exit 1
  done
done 2>&1 | grep -v amdgpu.ids
