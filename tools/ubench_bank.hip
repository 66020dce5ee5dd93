// Does the VGPR bank (register number mod 4) of the three sources of v_bitop3_b32
// matter on gfx950?  Four independent chains in v8..v11; the other two sources of
// each op sit in distinct banks ("spread"), in one other bank ("pair") or in the
// chain register's own bank ("same").  Also v_xor_b32 (VOP2) for reference.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bank.hip -o tools/ubench_bank.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v8", "v9", "v10", "v11", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

template <int MODE>
__global__ __launch_bounds__(64) void k_bank(unsigned *out, int iters)
{
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n"
                 "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n"
                 "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
                 :: "v"(seed) : CLOB);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0)      // spread: sources in three different banks
            asm volatile(REP16("v_bitop3_b32 v8, v8, v17, v18 bitop3:0x96\n v_bitop3_b32 v9, v9, v18, v19 bitop3:0x96\n"
                               "v_bitop3_b32 v10, v10, v19, v16 bitop3:0x96\n v_bitop3_b32 v11, v11, v16, v17 bitop3:0x96\n") ::: CLOB);
        else if (MODE == 1) // pair: the two extra sources share a bank (not the chain's)
            asm volatile(REP16("v_bitop3_b32 v8, v8, v17, v21 bitop3:0x96\n v_bitop3_b32 v9, v9, v18, v22 bitop3:0x96\n"
                               "v_bitop3_b32 v10, v10, v19, v23 bitop3:0x96\n v_bitop3_b32 v11, v11, v16, v20 bitop3:0x96\n") ::: CLOB);
        else if (MODE == 2) // same: all three sources in one bank
            asm volatile(REP16("v_bitop3_b32 v8, v8, v16, v20 bitop3:0x96\n v_bitop3_b32 v9, v9, v17, v21 bitop3:0x96\n"
                               "v_bitop3_b32 v10, v10, v18, v22 bitop3:0x96\n v_bitop3_b32 v11, v11, v19, v23 bitop3:0x96\n") ::: CLOB);
        else if (MODE == 3) // VOP2 xor, sources in different banks
            asm volatile(REP16("v_xor_b32 v8, v8, v17\n v_xor_b32 v9, v9, v18\n v_xor_b32 v10, v10, v19\n v_xor_b32 v11, v11, v16\n") ::: CLOB);
        else if (MODE == 4) // VOP2 xor, both sources in one bank
            asm volatile(REP16("v_xor_b32 v8, v8, v16\n v_xor_b32 v9, v9, v17\n v_xor_b32 v10, v10, v18\n v_xor_b32 v11, v11, v19\n") ::: CLOB);
        else                // bitop3 with a repeated source (two distinct registers)
            asm volatile(REP16("v_bitop3_b32 v8, v8, v17, v17 bitop3:0x96\n v_bitop3_b32 v9, v9, v18, v18 bitop3:0x96\n"
                               "v_bitop3_b32 v10, v10, v19, v19 bitop3:0x96\n v_bitop3_b32 v11, v11, v16, v16 bitop3:0x96\n") ::: CLOB);
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v11" : "=v"(r) :: CLOB);
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE>
static void run(unsigned *out, int waves, const char *name)
{
    const int iters = 2000, grid = 256 * 4 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_bank<MODE>), dim3(grid), dim3(64), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %d wave(s)/SIMD: %.3f ns per instruction per SIMD\n", name, waves, best * 1e6 / (iters * 64.0 * waves));
}

int main()
{
    unsigned *out;
    (void)hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(unsigned));
    for (int w = 1; w <= 4; w++) {
        run<0>(out, w, "bitop3, sources in 3 banks");
        run<1>(out, w, "bitop3, two sources share a bank");
        run<2>(out, w, "bitop3, all sources in one bank");
        run<5>(out, w, "bitop3, repeated source register");
        run<3>(out, w, "xor (VOP2), sources in 2 banks");
        run<4>(out, w, "xor (VOP2), sources in one bank");
    }
    return 0;
}
