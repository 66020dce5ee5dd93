#!/bin/bash
# Same-box A/B of the HOST-LAUNCHED bench path between the round-4 tree (a git worktree under tmp_r04/, built there) and this one,
# and the crossing: round 4's bench.py on THIS tree's library.   gpurun -- 'bash tools/ab_r04_host_launched.sh'
# Before:  git worktree add -f tmp_r04 8d9a0b8 && (cd tmp_r04 && python -m stereomatching_amd.build)   -- the worktree must lie
# inside the repository to travel with gpurun; remove it afterwards (git worktree remove --force tmp_r04), it is not tracked.
F="--no-graph --no-cpu-baseline --no-e2e --no-cost-modes"
q='import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"])'
for rep in 1 2 3; do
  for steps in 200 20; do
    a=$(python3 tmp_r04/bench.py $F --steps $steps --warmup 5 2>/dev/null | python3 -c "$q")
    x=$(SM_HIP_LIB=$PWD/stereomatching_amd/libstereo_hip.so python3 tmp_r04/bench.py $F --steps $steps --warmup 5 2>/dev/null | python3 -c "$q")
    b=$(python3 bench.py $F --no-c2 --steps $steps --warmup 5 2>/dev/null | python3 -c "$q")
    echo "--no-graph --steps $steps: round-4 tree $a | round-4 bench.py on round-5 library $x | round-5 tree $b   (ms per step, kernel ms)"
  done
done
