#!/bin/bash
# One box: the C3 (and 21x21) match launch under every combination of priority class, s_setprio policy and schedule,
# interleaved (tools/ab_variants.py; stereomatching_amd/variants/lib.so = the shipped library).
#   gpurun -- 'bash tools/prio_sweep.sh'   -> gpurun_out/prio_sweep.txt
E=""
for oc in 2 1; do for cl in 2 1; do for pat in 0xF0F0F0F0 0x0F0F0F0F 0xCCCCF0F0 0x33330F0F; do
  E="$E;SM_PRIO_ON_CHANGE=$oc,SM_PRIO_CLASS=$cl,SM_PATTERN=$pat"
done; done; done
E=${E#;}
{ tools/ubench_prio.bin 3 0 | head -1
  AB_ENVS="$E" timeout -k 10 400 python3 tools/ab_variants.py C3 1 ${1:-7} 2>&1 | grep -v amdgpu.ids | grep "^C3"
  AB_ENVS="SM_PRIO_ON_CHANGE=1;SM_PRIO_ON_CHANGE=1,SM_PRIO_CLASS=1;SM_PRIO_CLASS=1" timeout -k 10 400 python3 tools/ab_variants.py REF4K 1 ${1:-7} 2>&1 | grep -v amdgpu.ids | grep "^REF4K"
} > gpurun_out/prio_sweep.txt 2>&1
cat gpurun_out/prio_sweep.txt
