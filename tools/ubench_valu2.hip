// Issue-rate survey of candidate VALU instructions (gfx950), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define OPS(X) \
  X(0,  "v_and_b32 %0, %0, %1") \
  X(1,  "v_max_u32 %0, %0, %1") \
  X(2,  "v_min_u32 %0, %0, %1") \
  X(3,  "v_mul_u32_u24 %0, %0, %1") \
  X(4,  "v_mad_u32_u24 %0, %0, 16, %1") \
  X(5,  "v_and_or_b32 %0, %0, %1, %2") \
  X(6,  "v_or3_b32 %0, %0, %1, %2") \
  X(7,  "v_add3_u32 %0, %0, %1, %2") \
  X(8,  "v_lshl_add_u32 %0, %0, 4, %1") \
  X(9,  "v_add_lshl_u32 %0, %0, %1, 4") \
  X(10, "v_xad_u32 %0, %0, %1, %2") \
  X(11, "v_cndmask_b32 %0, %0, %1, vcc") \
  X(12, "v_cmp_gt_u32 vcc, %0, %1") \
  X(13, "v_bfi_b32 %0, %0, %1, %2") \
  X(14, "v_perm_b32 %0, %0, %1, %2") \
  X(15, "v_sad_u8 %0, %0, %1, %2") \
  X(16, "v_sad_u16 %0, %0, %1, %2") \
  X(17, "v_dot4_u32_u8 %0, %0, %1, %2") \
  X(18, "v_dot8_u32_u4 %0, %0, %1, %2") \
  X(19, "v_pk_add_u16 %0, %0, %1") \
  X(20, "v_pk_max_u16 %0, %0, %1") \
  X(21, "v_pk_lshlrev_b16 %0, 4, %0") \
  X(22, "v_pk_mad_u16 %0, %0, %1, %2") \
  X(23, "v_lshrrev_b32 %0, 3, %0") \
  X(24, "v_ashrrev_i32 %0, 3, %0") \
  X(25, "v_sub_u32 %0, %0, %1") \
  X(26, "v_or_b32 %0, %0, %1") \
  X(27, "v_mov_b32 %0, %1") \
  X(28, "v_max_i32 %0, %0, %1") \
  X(29, "v_med3_u32 %0, %0, %1, %2") \
  X(30, "v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0") \
  X(31, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
  X(32, "v_max_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(33, "v_msad_u8 %0, %0, %1, %2") \
  X(34, "v_mul_lo_u32 %0, %0, %1") \
  X(35, "v_pk_min_u16 %0, %0, %1") \
  X(36, "v_pk_sub_u16 %0, %0, %1") \
  X(37, "v_bfm_b32 %0, %0, %1") \
  X(38, "v_mbcnt_lo_u32_b32 %0, %0, %1") \
  X(39, "v_xnor_b32 %0, %0, %1") \
  X(40, "v_max3_u32 %0, %0, %1, %2") \
  X(41, "v_bcnt_u32_b32 %0, %1, %0") \
  X(42, "v_lshl_or_b32 %0, %0, 4, %1") \
  X(43, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8") \
  X(44, "v_pk_mul_lo_u16 %0, %0, %1") \
  X(45, "v_cvt_pk_u8_f32 %0, %0, %1, %2") \
  X(46, "v_add_co_u32 %0, vcc, %0, %1") \
  X(47, "v_cmp_ne_u32 vcc, 0, %0") \
  X(48, "v_qsad_pk_u16_u8 %3, %3, %1, %3") \
  X(49, "v_mqsad_pk_u16_u8 %3, %3, %1, %3")

template <int OP>
__global__ __launch_bounds__(256) void k(u32 *out, int iters, u32 seed)
{
    u32 r[16];
    unsigned long long q[8];
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    for (int i = 0; i < 8; i++) q[i] = seed * (i + 3) + threadIdx.x;
    u32 s = seed | 1, s2 = seed * 7 + 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
#define X(N, STR) if (OP == N) asm volatile(STR : "+v"(r[i]) : "v"(s), "v"(s2), "v"(q[i & 7]) : "vcc");
                OPS(X)
#undef X
            }
        }
    }
    u32 acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    for (int i = 0; i < 8; i++) acc ^= (u32)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static const char *NAMES[] = {
#define X(N, STR) STR,
    OPS(X)
#undef X
};

template <int OP>
static void run(u32 *d)
{
    const int iters = 1000, wps = 4;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, d, 10, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 12345u);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / ((double)iters * 64 * wps);
    printf("%6.3f ns/wave-instr/SIMD  (x%.2f of v_and)  %s\n", ns, ns / 1.015, NAMES[OP]);
}

template <int N> struct Loop { static void go(u32 *d) { Loop<N - 1>::go(d); run<N - 1>(d); } };
template <> struct Loop<0> { static void go(u32 *) {} };

int main()
{
    u32 *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    Loop<50>::go(d);
    return 0;
}
