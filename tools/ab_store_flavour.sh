# Same-device A/B of the result-store flavours (variants sc1 / plain / nt built with -DSM_BS_STORE=1/2/0) x the two lane merges.
export AB_ENVS="SM_LANE_MERGE=1"
for step in "" 1; do
  export AB_STEP=$step
  echo "== AB_STEP=${step:-0}"
  timeout -k 10 120 python tools/ab_variants.py C4 8 7 &&
  timeout -k 10 120 python tools/ab_variants.py C3 1 7 &&
  timeout -k 10 120 python tools/ab_variants.py REF4K 1 7 || exit 1
done
