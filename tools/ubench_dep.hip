// Dependent-issue latency of full-rate VALU ops on gfx950: a wave runs ILP
// independent chains of v_bitop3_b32 (each op depends on the previous op of its
// chain).  ns per instruction per SIMD vs ILP, at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dep.hip -o tools/ubench_dep.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ __launch_bounds__(64) void k_dep(unsigned *out, int iters)
{
    unsigned a[ILP];
    unsigned x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u;
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++)
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP>
static void run(unsigned *out, int waves_per_simd)
{
    const int iters = 2000;
    const int grid = 256 * 4 * waves_per_simd;      // one-wave workgroups
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_dep<ILP>), dim3(grid), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)iters * 16 * ILP * waves_per_simd;
    printf("ILP %2d, %d wave(s)/SIMD: %.3f ns per instruction per SIMD (%.3f per wave)\n", ILP, waves_per_simd,
           best * 1e6 / instr_per_simd, best * 1e6 / (iters * 16.0 * ILP));
}

int main()
{
    unsigned *out;
    (void)hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(unsigned));
    for (int w = 1; w <= 3; w++) {
        run<1>(out, w); run<2>(out, w); run<3>(out, w); run<4>(out, w); run<5>(out, w); run<6>(out, w); run<8>(out, w); run<12>(out, w);
    }
    return 0;
}
