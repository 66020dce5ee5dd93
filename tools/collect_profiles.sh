#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against, on the
# GPU box:   gpurun -- 'bash tools/collect_profiles.sh r01'
# Writes raw output under gpurun_out/prof_<tag>/; summarise on the build box with
#   python tools/summarise_profiles.py gpurun_out/prof_<tag> profiles/<tag>
# Kernel trace and PMC passes are separate runs (never combined with other
# trace domains), as the pool requires.
set -u
TAG=${1:-r01}
ARGS=${2:---steps 20 --warmup 5 --no-cpu-baseline --no-e2e}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"   # gpurun merges results into the build box's copy: delete that one too before a re-run
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/kt.log" 2>&1 || echo "kernel-trace run failed"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_$n" -- python3 bench.py $ARGS > "$OUT/pmc_$n.log" 2>&1 || echo "pmc pass $n failed"
done
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench failed"
ls "$OUT"
