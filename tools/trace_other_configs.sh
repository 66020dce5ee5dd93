#!/bin/bash
# Kernel traces (rocprofv3 --kernel-trace --stats, no counters) of bench.py on the BASELINE
# configurations other than the headline one:  gpurun -- 'bash tools/trace_other_configs.sh'
# -> gpurun_out/trace_<cfg>/ ; the stats CSVs are copied to profiles/<round>/ by hand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "C2 1" "C4 8" "C5 1" "REF4K 1"; do
  set -- $spec
  out=gpurun_out/trace_$1
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --config $1 --pairs $2 \
    --steps 50 --warmup 10 --no-cpu-baseline --no-e2e > $out/bench.log 2>&1 || echo "$1 failed"
  f=$(ls $out/kt/*/*_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $1 x $2"; [ -n "$f" ] && head -4 "$f"
done
