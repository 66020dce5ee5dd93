#!/bin/bash
# gpurun -- 'bash tools/fit_valu_model.sh'     (PMC passes only, one counter set, no trace domains
# beyond --kernel-trace)  ->  gpurun_out/valu_fit/<cfg>_<pairs>_<best>_<tile_h>/{meta.json,*.csv}
# then here:  python tools/fit_valu_model.py gpurun_out/valu_fit profiles/r03/valu_fit.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/valu_fit; rm -rf $OUT; mkdir -p $OUT
CASES=${CASES:-"C3:1:0 C3:1:1 C2:1:0 C4:8:0 C5:1:0 REF4K:1:0"}
for c in $CASES; do
  IFS=: read cfg pairs best <<< "$c"
  for duo in 0 1; do for th in 8 16 32; do      # both workgroup shapes; th = rows per wave
    d=$OUT/${cfg}_${pairs}_${best}_duo${duo}_$th; mkdir -p $d
    flag=""; [ "$best" = 1 ] && flag="--best"
    echo "case $cfg pairs=$pairs best=$best duo=$duo tile_h=$th"
    SM_DUO=$duo SM_TILE_H=$th timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace \
      --output-format csv -d $d/prof -- python3 tools/one_launch.py $cfg $pairs $flag --meta $d/meta.json \
      > $d/log.txt 2>&1 || { echo "failed: $d"; tail -5 $d/log.txt; exit 1; }
  done; done
done
echo done
