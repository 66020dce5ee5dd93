#!/bin/bash
# same-device A/B of the cost-mode kernels of the libraries under stereomatching_amd/variants/:
#   gpurun -- 'bash tools/ab_cost.sh'  -> gpurun_out/ab_cost.txt
for c in ssd sad; do
  for cfg in C5 C3; do
    AB_COST=$c timeout -k 10 200 python tools/ab_variants.py $cfg 1 9
  done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_cost.txt
