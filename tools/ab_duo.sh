#!/bin/bash
# Same-device A/B of the two workgroup shapes of the bit-sliced kernel (one wave / two waves
# with a shared warm-up) over the benchmark configurations; the library under
# stereomatching_amd/variants/ is the product build, SM_DUO=0 forces one-wave workgroups.
mkdir -p gpurun_out
export AB_DESCRIBE=1 AB_ENVS="SM_DUO=0"
out=${1:-gpurun_out/ab_duo.txt}
(
timeout -k 10 120 python tools/ab_variants.py C3 1 9 &&
timeout -k 10 120 python tools/ab_variants.py C4 8 9 &&
timeout -k 10 120 python tools/ab_variants.py C5 1 9 &&
timeout -k 10 120 python tools/ab_variants.py C2 1 9 &&
timeout -k 10 120 python tools/ab_variants.py C1 1 9 &&
timeout -k 10 120 python tools/ab_variants.py REF4K 1 9
) > "$out" 2>&1
rc=$?
grep -v amdgpu.ids "$out"
exit $rc
