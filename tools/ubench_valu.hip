// Microbenchmark: issue rate of the integer VALU ops the match kernel is made
// of, relative to v_fma_f32, at 1 / 2 / 4 waves per SIMD.  GPU tuning aid.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { OP_FMA, OP_BCNT, OP_BFE, OP_BFEI, OP_ALIGN, OP_MAX3, OP_AND, OP_LSHL, OP_BITOP3, OP_XOR, OP_LSHLOR, OP_ADD, OP_MIX, OP_N };
static const char *NAMES[] = {"v_fma_f32", "v_bcnt_u32_b32", "v_bfe_u32", "v_bfe_i32", "v_alignbit_b32", "v_max3_u32",
                              "v_and_b32", "v_lshlrev_b32", "v_bitop3_b32", "v_xor_b32", "v_lshl_or_b32", "v_add_u32", "mix(bfe,bcnt,bfei,lshl,bitop3,max3)"};

template <int OP>
__global__ __launch_bounds__(256) void k(u32 *out, int iters, u32 seed)
{
    u32 r[16];
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    u32 s = seed | 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(s));
                if (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(s));
                if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 18" : "+v"(r[i]));
                if (OP == OP_BFEI) asm volatile("v_bfe_i32 %0, %0, 3, 18" : "+v"(r[i]));
                if (OP == OP_ALIGN) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r[i]) : "v"(s));
                if (OP == OP_MAX3) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "v"(r[(i + 1) & 15]));
                if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
                if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 10, %0" : "+v"(r[i]));
                if (OP == OP_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, 5 bitop3:0xc8" : "+v"(r[i]) : "v"(s));
                if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
                if (OP == OP_LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 10, %1" : "+v"(r[i]) : "v"(s));
                if (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
            }
            if (OP == OP_MIX) {
                // the match kernel's per-(pixel,shift) sequence, 10 candidates + 1 tail
#pragma unroll
                for (int i = 0; i < 10; i++) {
                    u32 t, m;
                    asm volatile("v_bfe_u32 %0, %1, 3, 18" : "=v"(t) : "v"(s));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(t));
                    asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(m) : "v"(s));
                    asm volatile("v_lshlrev_b32 %0, 10, %1" : "=v"(t) : "v"(r[i]));
                    asm volatile("v_bitop3_b32 %0, %0, %1, 5 bitop3:0xc8" : "+v"(t) : "v"(m));
                    asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(r[15]) : "v"(t));
                    asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[14]) : "v"(t));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[13]) : "v"(t));
                }
            }
        }
    }
    u32 acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static void run(u32 *d, int waves_per_simd)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double per_wave = (double)iters * 64 * (OP == OP_MIX ? 80.0 / 64.0 : 1.0);
    const double ns_per_instr_simd = best * 1e6 / (per_wave * waves_per_simd);
    printf("%-40s waves/SIMD %d: %8.3f ms  -> %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n",
           NAMES[OP], waves_per_simd, best, ns_per_instr_simd, ns_per_instr_simd * 2.4);
}

template <int OP> static void all(u32 *d) { run<OP>(d, 1); run<OP>(d, 2); run<OP>(d, 4); }

int main()
{
    u32 *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    all<OP_FMA>(d); all<OP_BCNT>(d); all<OP_BFE>(d); all<OP_BFEI>(d); all<OP_ALIGN>(d); all<OP_MAX3>(d);
    all<OP_AND>(d); all<OP_LSHL>(d); all<OP_BITOP3>(d); all<OP_XOR>(d); all<OP_LSHLOR>(d); all<OP_ADD>(d); all<OP_MIX>(d);
    return 0;
}
