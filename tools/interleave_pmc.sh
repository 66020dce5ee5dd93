#!/bin/bash
# gpurun -- 'bash tools/interleave_pmc.sh "<counter set>" ...'  -> gpurun_out/ilv/*
# PMC passes only (one per argument), no other trace domains; each pass under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/ilv; rm -rf $OUT; mkdir -p $OUT
[ $# -eq 0 ] && set -- "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for set in "$@"; do
  n=$(echo $set | cut -d" " -f1)
  echo "pass $n"
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$n -- python3 tools/interleave_pmc.py C4 8 > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $OUT/$n.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/ilv/*/*/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "k_match_bs" in r["Kernel_Name"]]
    if not rows: continue
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in by.items():
        v.sort(); vals = [x for _, x in v]; n = len(vals) // 2
        a, b = vals[:n], vals[n:]
        print(f"{c:28s} back-to-back {sum(a)/len(a):14.1f}   behind fill {sum(b)/len(b):14.1f}")
PY
