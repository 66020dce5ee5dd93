for cfg in C5 C3 C2; do
  AB_COST=ssd AB_ENVS="SM_COST_KERNEL=0;SM_COST_KERNEL=2" timeout -k 10 200 python tools/ab_variants.py $cfg 1 9
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_ssd_mfma.txt
