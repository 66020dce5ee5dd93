#!/bin/bash
# PMC counters of the match launch for a list of settings, same box:
#   gpurun -- 'bash tools/pmc_compare.sh "C4:8 C3:1" "SM_LANE_MERGE=1 SM_LANE_MERGE=2"'  -> gpurun_out/pmc_compare.txt
# (PMC passes only, with --kernel-trace; two counter sets per setting)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
CASES=${1:-"C4:8 C3:1"}
SETTINGS=${2:-"SM_LANE_MERGE=1 SM_LANE_MERGE=2"}
OUT=gpurun_out/pmc_compare; rm -rf $OUT; mkdir -p $OUT
for c in $CASES; do
  IFS=: read cfg pairs <<< "$c"
  for s in $SETTINGS; do
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
               "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
      d=$OUT/${cfg}_${pairs}_${s//[=,]/_}_$i; mkdir -p $d; i=$((i+1))
      env ${s//,/ } timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d/prof -- \
        python3 tools/one_launch.py $cfg $pairs --launches 4 > $d/log.txt 2>&1 || { echo "failed: $d"; tail -5 $d/log.txt; exit 1; }
    done
  done
done
python3 - <<'PY' | tee gpurun_out/pmc_compare.txt
import csv, glob, os, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc_compare/*/prof/**/*counter_collection.csv", recursive=True)):
    case = f.split("/")[2].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_match_bs" in r["Kernel_Name"] and int(r["Grid_Size"]) > 64 * 4:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows[case][k] = sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0]     # skip the first launch
names = sorted({k for r in rows.values() for k in r})
for case, r in rows.items():
    print(case)
    for k in names:
        if k in r:
            print(f"   {k:24s} {r[k]:16.0f}")
PY
