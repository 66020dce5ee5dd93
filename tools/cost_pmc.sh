#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the cost-mode kernels:  gpurun -- 'bash tools/cost_pmc.sh TAG C3 sad [extra args]'
set -u
TAG=${1:-cost}; CFG=${2:-C3}; COST=${3:-sad}; shift 3 || true
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/cost_launch.py $CFG $COST --launches 20 "$@" > "$OUT/kt.log" 2>&1 || echo "kernel-trace run failed"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_$n" -- python3 tools/cost_launch.py $CFG $COST --launches 5 "$@" > "$OUT/pmc_$n.log" 2>&1 || echo "pmc pass $n failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sad" in row["Name"] or "k_cost" in row["Name"] or "k_ssd" in row["Name"]:
            res.setdefault("kernel_stats", []).append({k: row[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")})
cnt = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sad" in row["Kernel_Name"] or "k_ssd" in row["Kernel_Name"] or "k_cost" in row["Kernel_Name"]:
            cnt[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
res["counters_per_launch"] = {f"{k[0]}|{k[1]}": sum(v) / len(v) for k, v in sorted(cnt.items())}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
