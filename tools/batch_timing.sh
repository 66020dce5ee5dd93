#!/bin/bash
# stereopar-batch on a C4-like job: 8 distinct 1080p pairs x 8 repeats = 64 pairs, 64 shifts,
# 7x7, through the C host (pinned async transfers, narrow maps), on every visible GPU.  GPU box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/batch
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from stereomatching_amd.synth import make_pair, write_pgm
lines = []
for j in range(8):
    a, b = make_pair(1920, 1080, 64, seed=500 + j)
    write_pgm(f'gpurun_out/batch/l{j}.pgm', a); write_pgm(f'gpurun_out/batch/r{j}.pgm', b)
    lines.append(f'gpurun_out/batch/l{j}.pgm gpurun_out/batch/r{j}.pgm')
open('gpurun_out/batch/list.txt', 'w').write('\n'.join(lines) + '\n')
PY
for r in ${REPEATS:-8 64}; do
  for b in 1 4 8; do
    echo "repeat $r, pairs per launch $b: $(./timing/stereopar-batch -n 64 -b $b -r $r gpurun_out/batch/list.txt 0.15 7)"
  done
done
