// Does the size of the loop body matter?  The match kernel's row loop is ~16 KB of
// 8-byte VOP3 instructions.  Four independent v_bitop3 chains, loop bodies of
// 64 ... 4096 instructions, 2 waves per SIMD (one-wave workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ifetch.hip -o tools/ubench_ifetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define B4 "v_bitop3_b32 v8, v8, v17, v18 bitop3:0x96\n v_bitop3_b32 v9, v9, v18, v19 bitop3:0x96\n" \
           "v_bitop3_b32 v10, v10, v19, v16 bitop3:0x96\n v_bitop3_b32 v11, v11, v16, v17 bitop3:0x96\n"
#define X4 "v_xor_b32 v8, v8, v17\n v_xor_b32 v9, v9, v18\n v_xor_b32 v10, v10, v19\n v_xor_b32 v11, v11, v16\n"
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
#define CLOB "v8", "v9", "v10", "v11", "v16", "v17", "v18", "v19"

// SKEW: waves enter the loop body at different times (a wave-dependent sleep), so
// that the waves of a CU fetch DIFFERENT parts of the body at any moment, as the
// waves of a real kernel do, instead of marching through it in lockstep.
template <int BODY, bool VOP2, bool SKEW = false>
__global__ __launch_bounds__(64) void k_body(unsigned *out, int iters)
{
    if (SKEW) {
        const int n = (blockIdx.x * 37) % 61;
        for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(2);
    }
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n" :: "v"(seed) : CLOB);
    for (int it = 0; it < iters; it++) {
        if (VOP2) {
            if (BODY == 64) asm volatile(R16(X4) ::: CLOB);
            if (BODY == 1024) asm volatile(R256(X4) ::: CLOB);
            if (BODY == 4096) asm volatile(R1024(X4) ::: CLOB);
        } else {
            if (BODY == 64) asm volatile(R16(B4) ::: CLOB);
            if (BODY == 256) asm volatile(R64(B4) ::: CLOB);
            if (BODY == 1024) asm volatile(R256(B4) ::: CLOB);
            if (BODY == 2048) { asm volatile(R256(B4) ::: CLOB); asm volatile(R256(B4) ::: CLOB); }
            if (BODY == 4096) asm volatile(R1024(B4) ::: CLOB);
        }
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v11" : "=v"(r) :: CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int BODY, bool VOP2, bool SKEW = false>
static void run(unsigned *out, int waves)
{
    const int total = 1 << 18;              // instructions per wave
    const int iters = total / BODY, grid = 256 * 4 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_body<BODY, VOP2, SKEW>), dim3(grid), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%s%s body %4d instr (%5d B), %d wave(s)/SIMD: %.3f ns per instruction per SIMD\n", SKEW ? "skewed " : "", VOP2 ? "v_xor   " : "v_bitop3",
           BODY, BODY * (VOP2 ? 4 : 8), waves, best * 1e6 / ((double)iters * BODY * waves));
}

int main()
{
    unsigned *out;
    (void)hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(unsigned));
    for (int w = 1; w <= 2; w++) {
        run<64, false>(out, w); run<256, false>(out, w); run<1024, false>(out, w); run<2048, false>(out, w); run<4096, false>(out, w);
        run<64, true>(out, w); run<1024, true>(out, w); run<4096, true>(out, w);
        run<1024, false, true>(out, w); run<2048, false, true>(out, w); run<4096, false, true>(out, w); run<4096, true, true>(out, w);
    }
    return 0;
}
