#!/bin/bash
# Whole-pipeline wall times of the timing-build CLI programs on synthetic PGM
# pairs of the reference's test sizes (the equivalent of its test/time.sh,
# which needs the PNG test images and `bc`).  GPU box.
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from stereomatching_amd.synth import make_pair, write_pgm
import os
os.makedirs('gpurun_out/cli', exist_ok=True)
for w, h in ((240,135),(480,270),(960,540),(1920,1080),(3840,2160)):
    a, b = make_pair(w, h, 30, seed=w)
    write_pgm(f'gpurun_out/cli/a_{w}.pgm', a); write_pgm(f'gpurun_out/cli/b_{w}.pgm', b)
PY
for prog in stereopar stereopar-ghost; do
  for w in 240 480 960 1920 3840; do
    for rep in 1 2; do out=$(./timing/$prog gpurun_out/cli/a_$w.pgm gpurun_out/cli/b_$w.pgm); done
    echo "$prog $w default(S=21): $(echo $out | awk '{print $15}') s"
  done
  out=$(./timing/$prog gpurun_out/cli/a_3840.pgm gpurun_out/cli/b_3840.pgm 0.15 9); echo "$prog 3840 S=9: $(echo $out | awk '{print $15}') s"
  out=$(STEREO_NUM_SHIFTS=128 ./timing/$prog gpurun_out/cli/a_3840.pgm gpurun_out/cli/b_3840.pgm 0.15 9); echo "$prog 3840 S=9 D=128: $(echo $out | awk '{print $15}') s"
done
for w in 240 480; do out=$(STEREO_FAITHFUL=1 ./timing/stereomatch gpurun_out/cli/a_$w.pgm gpurun_out/cli/b_$w.pgm); echo "stereomatch(faithful oracle) $w default: $(echo $out | awk '{print $15}') s"; done
