#!/bin/bash
# stereopar-batch's two host threads per device under ThreadSanitizer (the C host is CPU code;
# the GPU side is untouched).  Build on the build box, run on the GPU box:
#   gcc -std=gnu11 -g -O1 -fsanitize=thread -Iinclude -Ioracle -DNO_WRITES -DSTEREOPAR_BATCH_TEST_HOOKS \
#       stereomatching_amd/host/stereopar_batch.c stereomatching_amd/host/image.c \
#       -o timing/stereopar-batch-tsan -Lstereomatching_amd -lstereo_hip \
#       -Wl,-rpath,'$ORIGIN/../stereomatching_amd' -lm -lpthread
#   gpurun -- 'bash tools/tsan_batch.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/batch
python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from stereomatching_amd.synth import make_pair, write_pgm
lines = []
for j in range(4):
    a, b = make_pair(640, 360, 64, seed=500 + j)
    write_pgm(f"gpurun_out/batch/sl{j}.pgm", a); write_pgm(f"gpurun_out/batch/sr{j}.pgm", b)
    lines.append(f"gpurun_out/batch/sl{j}.pgm gpurun_out/batch/sr{j}.pgm")
open("gpurun_out/batch/slist.txt", "w").write("\n".join(lines) + "\n")
PY
# TSan needs a fixed address-space layout on this kernel (setarch -R)
TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0" timeout -k 10 200 setarch x86_64 -R \
  ./timing/stereopar-batch-tsan -n 64 -b 2 -r 6 gpurun_out/batch/slist.txt 0.15 7 \
  > gpurun_out/tsan_out.txt 2> gpurun_out/tsan_err.txt
echo "success path: rc=$?"
cat gpurun_out/tsan_out.txt
echo "reports: $(grep -c 'WARNING: ThreadSanitizer' gpurun_out/tsan_err.txt)"
echo "racing accesses inside the program's own functions: $(grep -cE '^ +#[0-9]+ (collector_main|worker_main|set_failed|sum_bytes)' gpurun_out/tsan_err.txt)"
# the error path: an injected failure (-x) with batches in flight, two workers on device 0
TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0" timeout -k 10 200 setarch x86_64 -R \
  ./timing/stereopar-batch-tsan -d 0,0 -n 64 -b 1 -r 20 -x 5 gpurun_out/batch/slist.txt 0.15 7 \
  > gpurun_out/tsan_fail_out.txt 2> gpurun_out/tsan_fail_err.txt
echo "failure path (-x 5, -d 0,0): rc=$? (1 expected), message: $(grep -c 'injected failure' gpurun_out/tsan_fail_err.txt)"
echo "reports: $(grep -c 'WARNING: ThreadSanitizer' gpurun_out/tsan_fail_err.txt)"
echo "racing accesses inside the program's own functions: $(grep -cE '^ +#[0-9]+ (collector_main|worker_main|set_failed|sum_bytes)' gpurun_out/tsan_fail_err.txt)"
