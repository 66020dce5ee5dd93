#!/bin/bash
# One box, the REAL step (edges + match back to back, as bench.py runs it), one setting after the other, twice over.
# (A timing loop of match launches alone is NOT a stand-in for it: the match launch that follows the edge kernel finds
# its workgroups in other slots than one that follows itself, and priority settings rank differently.)
#   gpurun -- 'bash tools/sustained_ab.sh [cfg] [settings file]'  -> gpurun_out/sustained_ab.txt
CFG=${1:-C3}
SETTINGS=${2:-tools/sustained_ab.settings}
{ for rep in 1 2; do
  while IFS= read -r s; do
    echo "== rep $rep: ${s:-shipped}"
    env $s timeout -k 10 100 python3 tools/sustained.py $CFG 2 ${STEPS:-10000} 2>&1 | grep "^t ="
  done < "$SETTINGS"
done; } > gpurun_out/sustained_ab.txt 2>&1
cat gpurun_out/sustained_ab.txt
