#!/bin/bash
# bench.py over the BASELINE configurations on one GPU (one gpurun call, same device):
#   gpurun -- 'bash tools/bench_all_configs.sh'   -> gpurun_out/bench_all.jsonl
# The single pairs are run a second time with --overlap (the steps issued through sm_run_after, whose only input
# dependency is an event: the plan then runs consecutive steps on its two lanes, also inside the HIP graph).
mkdir -p gpurun_out
: > gpurun_out/bench_all.jsonl
for spec in "C1 1" "C1 1 --overlap" "C2 1" "C2 1 --overlap" "C4 8" "C3 1" "C3 1 --overlap" "C3 8" "C5 1" "C5 1 --overlap" "REF4K 1"; do
  set -- $spec
  python3 bench.py --config $1 --pairs $2 $3 --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-cost-modes --no-c2 2>/dev/null | tail -1 >> gpurun_out/bench_all.jsonl || exit 1
done
python3 - <<'PY'
import json
for l in open("gpurun_out/bench_all.jsonl"):
    d = json.loads(l)
    print(f'{d["config"]["workload"][:60]:60s} {"overlapped" if d["config"]["pipelined"] else "          "} {"verified" if d["verified"] else "NOT VERIFIED"} step {d["ms_per_step"]:.4f} ms  match {d["roofline"]["kernel_ms"]:.4f} ms  {d["value"]/1e6:.2f} M')
PY
