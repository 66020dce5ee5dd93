#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of one C3 (or given) match launch, both lane merges: gpurun -- 'bash tools/write_size_check.sh [C3:1]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
C=${1:-C3:1}; IFS=: read cfg pairs <<< "$C"
for s in SM_LANE_MERGE=2 SM_LANE_MERGE=1; do
  for set in WRITE_SIZE FETCH_SIZE; do
    d=gpurun_out/wsize/${cfg}_${s//=/_}_$set; rm -rf $d; mkdir -p $d
    env $s timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d/prof -- python3 tools/one_launch.py $cfg $pairs --launches 3 > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
    python3 - "$d" "$s" "$set" <<'PY'
import csv, glob, sys
d, s, name = sys.argv[1:]
for f in glob.glob(d + "/prof/**/*counter_collection.csv", recursive=True):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_match_bs" in r["Kernel_Name"] and int(r["Grid_Size"]) > 256]
    print(f"{s:18s} {name:11s} {sum(v)/len(v):12.1f} KiB per launch ({len(v)} launches)")
PY
  done
done
