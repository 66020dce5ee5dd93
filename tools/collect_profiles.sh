#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against, on the GPU box:
#     gpurun -- 'bash tools/collect_profiles.sh r05'
# Writes raw output under gpurun_out/prof_<tag>/; summarise on the build box with
#     python tools/summarise_profiles.py gpurun_out/prof_<tag> profiles/<tag>
# Kernel trace and PMC passes are separate runs (never combined with other trace domains), as the pool
# requires.  Order: unprofiled runs first (same box, same clocks), then the trace, then the counters.
#
# bench.py replays its steps from a HIP graph (its default; 2000 steps in the traced run): under the tracer a
# launch costs the host more than the 0.09 ms a step takes, the traced process of round 3 was host-bound
# (0.115 ms per step) and its kernels ran 13 % slower on the sagging clocks than the bench line next to
# them.  Replayed from a graph the traced process is GPU-bound like the untraced one, and the trace's
# average agrees with the untraced kernel time (bench_traced.json is the traced process's own line).
set -u
TAG=${1:-r05}
F="--no-cpu-baseline --no-e2e --no-cost-modes --no-c2 --no-overlapped"
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"   # gpurun merges results into the build box's copy: delete that one too before a re-run
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || echo "bench failed"
python3 bench.py --steps 20 --warmup 5 $F > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err" || echo "bench (driver flags) failed"
python3 bench.py --no-graph $F > "$OUT/bench_nograph.json" 2> "$OUT/bench_nograph.err" || echo "bench --no-graph failed"
python3 bench.py --no-graph --steps 20 --warmup 5 $F > "$OUT/bench_nograph_driver_flags.json" 2> "$OUT/bench_nograph_driver_flags.err" || echo "bench --no-graph (driver flags) failed"
python3 bench.py --steps 2000 $F > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err" || echo "bench --steps 2000 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 2000 $F > "$OUT/bench_traced.json" 2> "$OUT/kt.log" || echo "kernel-trace run failed"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_$n" -- python3 bench.py --steps 20 --warmup 5 $F > "$OUT/pmc_$n.log" 2>&1 || echo "pmc pass $n failed"
done
python3 bench.py --steps 200 $F > "$OUT/bench_after.json" 2> "$OUT/bench_after.err" || echo "bench (after) failed"
ls "$OUT"
