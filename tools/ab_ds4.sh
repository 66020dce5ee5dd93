#!/bin/bash
# 4 shifts per lane on the small configurations, forced through the plan options, against the plan's own choice
export AB_DESCRIBE=1
export AB_ENVS="SM_DS=4;SM_DS=4,SM_LANE_MERGE=2;SM_DS=4,SM_LANE_MERGE=2,SM_TILE_H=4;SM_DS=4,SM_LANE_MERGE=2,SM_TILE_H=8;SM_DS=4,SM_DUO=0;SM_DS=4,SM_DUO=0,SM_LANE_MERGE=2,SM_TILE_H=4;SM_DS=4,SM_TILE_H=2;SM_DS=4,SM_TILE_H=3"
for step in "" 1; do
  export AB_STEP=$step
  echo "== AB_STEP=${step:-0}"
  timeout -k 10 200 python tools/ab_variants.py C1 1 9
  timeout -k 10 200 python tools/ab_variants.py C2 1 9
done 2>&1 | grep -v amdgpu.ids
