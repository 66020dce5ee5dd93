#!/bin/bash
# SSD, same device: the libraries under stereomatching_amd/variants/ -- by default a copy of the product library, whose
# plan takes the matrix-core kernel (k_ssd_mfma); with cost_kernel = 2 the byte-dot kernel (k_ssd_dot)
#   gpurun -- 'bash tools/ab_ssd_kernels.sh'  -> gpurun_out/ab_ssd_kernels.txt
for cfg in C5 C3 C2; do
  AB_COST=ssd AB_ENVS="SM_COST_KERNEL=2" timeout -k 10 200 python tools/ab_variants.py $cfg 1 9
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_ssd_kernels.txt
