// VALU issue rate on gfx950 under CONTROLLED occupancy: exactly W waves on every SIMD
// (W = 1: the kernel claims > 256 VGPR + AGPR; W = 2: 176..256), checked with a HW_ID
// census -- one-wave workgroups of small kernels are otherwise stacked unevenly by the
// dispatcher and "waves per SIMD" is not what the grid size suggests.
// A wave runs ILP independent chains round-robin, so an instruction depends on the one
// issued ILP instructions earlier (dependency distance = ILP).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

enum { OP_BITOP3, OP_XOR, OP_MIXED };

template <int ILP, int OP, int W>
__global__ __launch_bounds__(64) void k_issue(unsigned *out, unsigned long long *info, int iters)
{
    if (W == 1) asm volatile("" ::: "v250", "a16");     // > 256 registers: one wave per SIMD
    if (W == 2) asm volatile("" ::: "v200");            // 201..256: two waves per SIMD
    unsigned a[ILP];
    unsigned x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u, z = x ^ 0x5bd1e995u;
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 24; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (OP == OP_BITOP3 || (OP == OP_MIXED && ((r * ILP + i) % 8) != 7))
                    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(x), "v"(y));
                else if (OP == OP_XOR)
                    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
                else
                    asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(z));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        info[blockIdx.x * 2] = t1 - t0;
        info[blockIdx.x * 2 + 1] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |
                                   (unsigned)__builtin_amdgcn_s_getreg(63492);
    }
}

template <int ILP, int OP, int W>
static void run(unsigned *out, unsigned long long *info, const char *name)
{
    const int iters = 400;
    const int grid = 256 * 4 * W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_issue<ILP, OP, W>), dim3(grid), dim3(64), 0, 0, out, info, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(grid * 2);
    (void)hipMemcpy(h.data(), info, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> per_simd;
    double cyc = 0;
    for (int b = 0; b < grid; b++) {
        const unsigned long long hw = h[b * 2 + 1];
        const unsigned id = (unsigned)hw;
        // XCC | SE | SH | CU | SIMD
        per_simd[((hw >> 32) << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 11) | (((id >> 8) & 15) << 4) | ((id >> 4) & 3)]++;
        cyc += (double)h[b * 2];
    }
    int lo = 1 << 30, hi = 0;
    for (auto &kv : per_simd) { lo = kv.second < lo ? kv.second : lo; hi = kv.second > hi ? kv.second : hi; }
    const double n_instr = (double)iters * 24 * ILP;
    printf("%-8s dist %2d, %d wave(s)/SIMD [census: %4zu SIMDs, %d..%d waves each]: %.3f ns/instr/SIMD, "
           "%.2f shader cycles/instr/wave\n", name, ILP, W, per_simd.size(), lo, hi,
           best * 1e6 / (n_instr * W), cyc / grid / n_instr);
}

template <int OP, int W> static void sweep(unsigned *out, unsigned long long *info, const char *name)
{
    run<1, OP, W>(out, info, name); run<2, OP, W>(out, info, name); run<3, OP, W>(out, info, name);
    run<4, OP, W>(out, info, name); run<6, OP, W>(out, info, name); run<8, OP, W>(out, info, name);
    run<16, OP, W>(out, info, name);
}

int main()
{
    unsigned *out; unsigned long long *info;
    (void)hipMalloc(&out, 256 * 4 * 2 * 64 * sizeof(unsigned));
    (void)hipMalloc(&info, 256 * 4 * 2 * 2 * sizeof(unsigned long long));
    sweep<OP_BITOP3, 1>(out, info, "bitop3"); sweep<OP_BITOP3, 2>(out, info, "bitop3");
    sweep<OP_XOR, 1>(out, info, "xor");       sweep<OP_XOR, 2>(out, info, "xor");
    sweep<OP_MIXED, 2>(out, info, "7:1 mix");
    return 0;
}
