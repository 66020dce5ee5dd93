#!/bin/bash
# the four cost-mode cases of bench.py's extras through tools/cost_pmc.sh -> gpurun_out/cost_<cost>_<cfg>.json
# (copy them to profiles/<round>/ and run tools/make_cost_valu.py profiles/<round>)
for cc in "C5 ssd" "C3 ssd" "C5 sad" "C3 sad"; do
  set -- $cc
  bash tools/cost_pmc.sh ${2}_$1 $1 $2 > gpurun_out/cost_pmc_${2}_$1.log 2>&1
  cp gpurun_out/prof_${2}_$1/summary.json gpurun_out/cost_${2}_$1.json
done
ls gpurun_out/cost_*.json
