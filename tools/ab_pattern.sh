#!/bin/bash
# Same-device A/B of time-sliced priority schedules (SM_PATTERN, hex; bit k = which wave-slot
# parity is favoured during the k-th 16384-cycle unit of a wave's life)
mkdir -p gpurun_out
out=${1:-gpurun_out/ab_pattern.txt}
cfgs=${CFGS:-"C3:1 C4:8 C5:1"}
pats=${PATS:-"F0F0F0F0 1F8 F8 FC 3F0 3F8 FFFFFC3C FFFFFE07 0"}
envs=""
for p in $pats; do envs="$envs;SM_PATTERN=$p"; done
export AB_ENVS="${envs#;}"
(
for c in $cfgs; do IFS=: read cfg pairs <<< "$c"
  timeout -k 10 200 python tools/ab_variants.py $cfg $pairs 7 || exit 1
done
) > "$out" 2>&1
rc=$?
grep -v amdgpu.ids "$out"
exit $rc
