// Which stream / event pattern of a "pipelined" two-lane call sequence this HIP runtime accepts inside a
// stream capture, which it refuses with an error, and which one takes the process down.  Round 4 saw a
// segmentation fault when bench.py captured sm_run of a pipelined plan into a graph (gpurun_out/pl.err,
// profiles/r04/ab_prio_unit.txt); this probe isolates the suspects, ONE PER CHILD PROCESS (the parent forks
// before it touches HIP, so a crash is the child's and is reported as the signal that ended it).
//   hipcc --offload-arch=gfx950 -O2 tools/capture_probe.hip -o tools/capture_probe.bin && ./tools/capture_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

__global__ void k_add(int *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, v); }

#define TRY(call)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            printf("    %s -> %s\n", #call, hipGetErrorName(e_));                                   \
            (void)hipGetLastError();                                                                \
            failed = 1;                                                                             \
        }                                                                                           \
    } while (0)

static int scenario(int s, hipStreamCaptureMode mode)
{
    int failed = 0;
    hipStream_t user, lane[2];
    hipEvent_t ev_in, ev_free[4];
    int *d;
    TRY(hipSetDevice(0));
    TRY(hipStreamCreateWithFlags(&user, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) TRY(hipStreamCreateWithFlags(&lane[i], hipStreamNonBlocking));
    TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) TRY(hipEventCreateWithFlags(&ev_free[i], hipEventDisableTiming));
    TRY(hipMalloc(&d, 4));
    TRY(hipMemset(d, 0, 4));
    TRY(hipDeviceSynchronize());
    int expect = 0;

    if (s == 2 || s == 3 || s == 6) {           // events that were recorded eagerly before the capture
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[0], d, 1);
        TRY(hipEventRecord(ev_free[1], lane[0]));
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[1], d, 1);
        TRY(hipEventRecord(ev_free[2], lane[1]));
        TRY(hipDeviceSynchronize());
        expect += 2;
    }
    hipGraph_t graph = nullptr;
    TRY(hipStreamBeginCapture(user, mode));
    switch (s) {
    case 1:     // legal fork / join, events never recorded before
    case 2:     // the same, the join event had an eager record before the capture
        TRY(hipEventRecord(ev_in, user));
        TRY(hipStreamWaitEvent(lane[0], ev_in, 0));
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[0], d, 10);
        TRY(hipEventRecord(ev_free[1], lane[0]));
        TRY(hipStreamWaitEvent(user, ev_free[1], 0));
        expect += 10;
        break;
    case 3:     // a forked lane waits for an event whose only record is eager, from before the capture
        TRY(hipEventRecord(ev_in, user));
        TRY(hipStreamWaitEvent(lane[0], ev_in, 0));
        TRY(hipStreamWaitEvent(lane[0], ev_free[2], 0));
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[0], d, 10);
        TRY(hipEventRecord(ev_free[3], lane[0]));
        TRY(hipStreamWaitEvent(user, ev_free[3], 0));
        expect += 10;
        break;
    case 4:     // the lane never joins the capture: eager launch + eager record, the capturing stream waits for it
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[0], d, 100);
        TRY(hipEventRecord(ev_free[1], lane[0]));
        TRY(hipStreamWaitEvent(user, ev_free[1], 0));
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, user, d, 10);
        expect += 10;           // (+100 eagerly, once)
        break;
    case 5:     // a forked lane is not joined when the capture ends
        TRY(hipEventRecord(ev_in, user));
        TRY(hipStreamWaitEvent(lane[0], ev_in, 0));
        hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, lane[0], d, 10);
        expect += 10;
        break;
    case 6: {   // the round-4 sequence in ordered mode: four calls on alternating lanes, call q waits for q - 3
        int set[4] = {0, 1, 1, 0};
        for (unsigned q = 3; q < 7; q++) {
            hipStream_t l = lane[q & 1];
            TRY(hipEventRecord(ev_in, user));
            TRY(hipStreamWaitEvent(l, ev_in, 0));
            if (set[(q - 3) & 3]) TRY(hipStreamWaitEvent(l, ev_free[(q - 3) & 3], 0));
            hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, l, d, 10);
            TRY(hipEventRecord(ev_free[q & 3], l));
            set[q & 3] = 1;
            TRY(hipStreamWaitEvent(user, ev_free[q & 3], 0));
            expect += 10;
        }
        break;
    }
    case 7: {   // overlap kept inside the graph: the fork event of call q + 1 is recorded BEFORE call q is joined
        hipEvent_t ev_fork[2];
        for (int i = 0; i < 2; i++) TRY(hipEventCreateWithFlags(&ev_fork[i], hipEventDisableTiming));
        TRY(hipEventRecord(ev_fork[1], user));
        for (unsigned q = 1; q < 5; q++) {
            hipStream_t l = lane[q & 1];
            TRY(hipStreamWaitEvent(l, ev_fork[q & 1], 0));
            hipLaunchKernelGGL(k_add, dim3(1), dim3(64), 0, l, d, 10);
            TRY(hipEventRecord(ev_free[q & 3], l));
            TRY(hipEventRecord(ev_fork[(q + 1) & 1], user));
            TRY(hipStreamWaitEvent(user, ev_free[q & 3], 0));
            expect += 10;
        }
        break;
    }
    }
    hipError_t ec = hipStreamEndCapture(user, &graph);
    printf("    hipStreamEndCapture -> %s\n", hipGetErrorName(ec));
    if (ec == hipSuccess && graph) {
        hipGraphExec_t exec = nullptr;
        size_t nodes = 0;
        TRY(hipGraphGetNodes(graph, nullptr, &nodes));
        TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        if (exec) {
            TRY(hipGraphLaunch(exec, user));
            TRY(hipStreamSynchronize(user));
        }
        int h = -1;
        TRY(hipDeviceSynchronize());
        TRY(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
        printf("    graph of %zu nodes replayed once: counter %d, expected %d%s\n", nodes, h, expect + (s == 4 ? 100 : 0),
               h == expect + (s == 4 ? 100 : 0) ? "" : "  <-- differs");
    } else {
        (void)hipGetLastError();
        failed = 1;
    }
    return failed;
}

int main(int argc, char **argv)
{
    static const char *what[] = {"", "fork / join with events recorded inside the capture only",
                                 "fork / join, the join event also had an eager record before the capture",
                                 "a forked lane waits for an event recorded eagerly BEFORE the capture",
                                 "the lane stays outside the capture; the capturing stream waits for its eager event",
                                 "a forked lane is left unjoined at hipStreamEndCapture",
                                 "round 4's ordered two-lane sequence (call q waits for call q - 3, first waits are pre-capture events)",
                                 "two-lane sequence with the next call's fork event recorded before the join (overlap inside the graph)"};
    for (int m = 0; m < 2; m++) {
        const hipStreamCaptureMode mode = m ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal;
        for (int s = 1; s <= 7; s++) {
            printf("[%s] scenario %d: %s\n", m ? "global" : "thread-local", s, what[s]);
            fflush(stdout);
            const pid_t pid = fork();           // (the parent has not touched HIP)
            if (pid == 0) {
                alarm(60);
                const int f = scenario(s, mode);
                fflush(stdout);
                _exit(f ? 3 : 0);
            }
            int st = 0;
            waitpid(pid, &st, 0);
            if (WIFSIGNALED(st)) printf("    => child ended by signal %d (%s)\n", WTERMSIG(st), strsignal(WTERMSIG(st)));
            else printf("    => %s\n", WEXITSTATUS(st) == 0 ? "accepted" : "refused with an error (no crash)");
            fflush(stdout);
        }
    }
    return 0;
}
