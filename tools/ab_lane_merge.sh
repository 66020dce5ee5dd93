#!/bin/bash
# Same-device A/B of the two lane merges of the bit-sliced kernel (through LDS every four rows / per row with
# DPP) over the benchmark configurations; the library under stereomatching_amd/variants/ is the product build,
# SM_LANE_MERGE=1 forces the per-row merge.  AB_STEP=1: the real step (edges, then match).
mkdir -p gpurun_out
export AB_DESCRIBE=1 AB_ENVS="SM_LANE_MERGE=1"
out=${1:-gpurun_out/ab_lane_merge.txt}
(
for step in "" 1; do
  export AB_STEP=$step
  echo "== AB_STEP=${step:-0} (0: match launches back to back; 1: edges + match)"
  timeout -k 10 120 python tools/ab_variants.py C3 1 9 &&
  timeout -k 10 120 python tools/ab_variants.py C4 8 9 &&
  timeout -k 10 120 python tools/ab_variants.py C5 1 9 &&
  timeout -k 10 120 python tools/ab_variants.py C2 1 9 &&
  timeout -k 10 120 python tools/ab_variants.py C1 1 9 &&
  timeout -k 10 120 python tools/ab_variants.py REF4K 1 9 || exit 1
done
) > "$out" 2>&1
rc=$?
grep -v amdgpu.ids "$out"
exit $rc
