#!/bin/bash
# bench.py over the BASELINE configurations on one GPU (one gpurun call, same device):
#   gpurun -- 'bash tools/bench_all_configs.sh'   -> gpurun_out/bench_all.jsonl
# The single small pairs (C1, C2) are run a second time with --pipeline (sm_plan_set_pipelined(1): the caller's
# promise that its inputs are complete at call time lets consecutive calls overlap).
mkdir -p gpurun_out
: > gpurun_out/bench_all.jsonl
for spec in "C1 1" "C1 1 --pipeline" "C2 1" "C2 1 --pipeline" "C4 8" "C3 1" "C3 8" "C5 1" "REF4K 1"; do
  set -- $spec
  python3 bench.py --config $1 --pairs $2 $3 --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-cost-modes 2>/dev/null | tail -1 >> gpurun_out/bench_all.jsonl || exit 1
done
python3 - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(l)
    print(f'{d["config"]["workload"][:60]:60s} {"pipelined" if d["config"]["pipelined"] else "         "} step {d["ms_per_step"]:.4f} ms  match {d["roofline"]["kernel_ms"]:.4f} ms  {d["value"]/1e6:.2f} M')
PY
