#!/bin/bash
# the ghost-border strip by k_cost_strip (default) against the general masked kernel (cost_kernel = 3), C5, same device:
#   gpurun -- 'bash tools/ab_cost_strip.sh'  -> gpurun_out/ab_cost_strip.txt
for c in ssd sad; do
  AB_COST=$c AB_ENVS="SM_COST_KERNEL=3" timeout -k 10 200 python tools/ab_variants.py C5 1 9
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_cost_strip.txt
