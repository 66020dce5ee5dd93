// What one (pixel, 4 shifts) row step of a SAD block matcher costs on gfx950, by the mix of
// instructions it is built from.  Round 5: the review proposed to rebuild k_sad_qs around column
// sums (2 v_qsad per step instead of 6); every such scheme trades quarter-rate v_qsad for packed
// adds / lane exchanges, so what decides is the price of those next to a v_qsad -- measured here
// under controlled occupancy (W waves on every SIMD, as tools/ubench_sad.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sadmix.hip -o tools/ubench_sadmix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32;

enum { M_TODAY, M_TODAY_SUB32, M_PREFIX, M_PREFIX_PK, M_ONCE_DPP, M_KEYS, M_QSAD6, M_QSAD4, M_QSAD2, M_ADD64, M_COUNT };
static const char *mix_name[M_COUNT] = {
    "today: 6 qsad + 2 pk_sub + 6 key ops",
    "today with v_sub_u32: 6 qsad + 2 sub32 + 6 key ops",
    "prefix chains: 4 qsad + 2 add32 + 2 sub32 + 6 key ops",
    "prefix chains, packed: 4 qsad + 2 pk_add + 2 pk_sub + 6 key ops",
    "each group once: 2 qsad + 2 sub32 + 4 dpp mov + 4 add/sub32 + 6 key ops",
    "6 key ops alone (2 lshl_or + 2 bitop3 + 2 min3)",
    "6 qsad alone", "4 qsad alone", "2 qsad alone",
    "v_lshl_add_u64 x 8",
};
static const int mix_instr[M_COUNT] = {14, 14, 14, 14, 18, 6, 6, 4, 2, 8};

#define QSAD(acc, r8, l4) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(r8), "v"(l4))
#define KEYS(Al, Ah, run)                                                                                  \
    do {                                                                                                   \
        u32 k0, k1, k2, k3;                                                                                \
        asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(k0) : "v"(Al), "v"(y));                         \
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(k1) : "v"(Al), "v"(x), "v"(y));      \
        asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(k2) : "v"(Ah), "v"(y));                         \
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(k3) : "v"(Ah), "v"(x), "v"(y));      \
        asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(run) : "v"(k0), "v"(k1));                          \
        asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(run) : "v"(k2), "v"(k3));                          \
    } while (0)

template <int MIX, int W>
__global__ __launch_bounds__(64) void k_mix(u32 *out, u64 *info, int iters)
{
    if (W == 1) asm volatile("" ::: "v250", "a16");
    if (W == 2) asm volatile("" ::: "v200");
    if (W == 3) asm volatile("" ::: "v160");
    if (W == 4) asm volatile("" ::: "v120");
    constexpr int NA = 8;
    u64 a[NA];
    const u32 x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u;
    const u64 xx = ((u64)x << 32) | (x ^ 0x5bd1e995u);
#pragma unroll
    for (int i = 0; i < NA; i++) a[i] = xx + i;
    u64 P = xx ^ 77, Q = xx ^ 99;
    u32 run0 = x, run1 = y;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 2; i++) {               // two steps side by side, as the kernel's two pixels
                u64 &A = a[i];
                u32 &Al = reinterpret_cast<u32 *>(&A)[0], &Ah = reinterpret_cast<u32 *>(&A)[1];
                u32 &run = i ? run1 : run0;
                if (MIX == M_TODAY || MIX == M_TODAY_SUB32) {
                    u64 t = 0;
                    u32 &tl = reinterpret_cast<u32 *>(&t)[0], &th = reinterpret_cast<u32 *>(&t)[1];
                    QSAD(t, xx, y); QSAD(t, a[7], x); QSAD(t, a[6], y);
                    QSAD(A, xx, x); QSAD(A, a[7], y); QSAD(A, a[6], x);
                    if (MIX == M_TODAY) {
                        asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Al) : "v"(tl));
                        asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Ah) : "v"(th));
                    } else {
                        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Al) : "v"(tl));
                        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Ah) : "v"(th));
                    }
                    KEYS(Al, Ah, run);
                }
                if (MIX == M_PREFIX || MIX == M_PREFIX_PK) {
                    u32 &Pl = reinterpret_cast<u32 *>(&P)[0], &Ph = reinterpret_cast<u32 *>(&P)[1];
                    u32 &Ql = reinterpret_cast<u32 *>(&Q)[0], &Qh = reinterpret_cast<u32 *>(&Q)[1];
                    QSAD(P, xx, y); QSAD(P, a[7], x);
                    QSAD(Q, a[6], y); QSAD(Q, a[5], x);
                    if (MIX == M_PREFIX) {
                        asm volatile("v_add_u32 %0, %0, %1" : "+v"(Al) : "v"(Pl));
                        asm volatile("v_add_u32 %0, %0, %1" : "+v"(Ah) : "v"(Ph));
                        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Al) : "v"(Ql));
                        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Ah) : "v"(Qh));
                    } else {
                        asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(Al) : "v"(Pl));
                        asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(Ah) : "v"(Ph));
                        asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Al) : "v"(Ql));
                        asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Ah) : "v"(Qh));
                    }
                    KEYS(Al, Ah, run);
                }
                if (MIX == M_ONCE_DPP) {
                    u64 t = 0, &G = a[2 + i], n;
                    u32 &tl = reinterpret_cast<u32 *>(&t)[0], &th = reinterpret_cast<u32 *>(&t)[1];
                    u32 &Gl = reinterpret_cast<u32 *>(&G)[0], &Gh = reinterpret_cast<u32 *>(&G)[1];
                    u32 &nl = reinterpret_cast<u32 *>(&n)[0], &nh = reinterpret_cast<u32 *>(&n)[1];
                    QSAD(G, xx, y); QSAD(t, a[7], x);
                    asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Gl) : "v"(tl));
                    asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Gh) : "v"(th));
                    // the neighbour class's group sum: lanes 1..3 of a quad from the lane below, lane 0 from lane 3 of another register
                    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xe" : "=v"(nl) : "v"(Gl));
                    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0x1" : "+v"(nl) : "v"(tl));
                    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xe" : "=v"(nh) : "v"(Gh));
                    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0x1" : "+v"(nh) : "v"(th));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(Al) : "v"(nl));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(Ah) : "v"(nh));
                    asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Al) : "v"(Gl));
                    asm volatile("v_sub_u32 %0, %0, %1" : "+v"(Ah) : "v"(Gh));
                    KEYS(Al, Ah, run);
                }
                if (MIX == M_KEYS) KEYS(Al, Ah, run);
                if (MIX == M_QSAD6) { QSAD(A, xx, y); QSAD(A, a[7], x); QSAD(A, a[6], y); QSAD(a[2 + i], xx, x); QSAD(a[2 + i], a[7], y); QSAD(a[2 + i], a[6], x); }
                if (MIX == M_QSAD4) { QSAD(A, xx, y); QSAD(A, a[7], x); QSAD(a[2 + i], xx, x); QSAD(a[2 + i], a[7], y); }
                if (MIX == M_QSAD2) { QSAD(A, xx, y); QSAD(a[2 + i], xx, x); }
                if (MIX == M_ADD64) {
#pragma unroll
                    for (int k = 0; k < 8; k++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[(k + i) & 3]) : "v"(xx));
                }
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 s = run0 ^ run1 ^ (u32)P ^ (u32)(P >> 32) ^ (u32)Q ^ (u32)(Q >> 32);
#pragma unroll
    for (int i = 0; i < NA; i++) s ^= (u32)a[i] ^ (u32)(a[i] >> 32);
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) info[blockIdx.x] = t1 - t0;
}

static int g_iters = 200;
template <int MIX, int W>
static void run(u32 *out, u64 *info)
{
    const int grid = 256 * 4 * W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<MIX, W>), dim3(grid), dim3(64), 0, 0, out, info, g_iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<u64> h(grid);
    (void)hipMemcpy(h.data(), info, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < grid; b++) cyc += (double)h[b];
    const double steps = (double)g_iters * 16 * 2;          // (pixel, 4 shifts) row steps per wave
    // wall time of one step per SIMD: W waves share the SIMD
    printf("%-72s %d wave(s)/SIMD: %6.2f ns per step and SIMD  (%2d instr, %.2f ns each; %.1f shader cycles per step and wave)\n",
           mix_name[MIX], W, best * 1e6 / (steps * W), mix_instr[MIX], best * 1e6 / (steps * W) / mix_instr[MIX], cyc / grid / steps);
}

template <int MIX> static void all(u32 *out, u64 *info) { run<MIX, 1>(out, info); run<MIX, 2>(out, info); run<MIX, 3>(out, info); }

int main()
{
    u32 *out; u64 *info;
    (void)hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(u32));
    (void)hipMalloc(&info, 256 * 4 * 4 * sizeof(u64));
    all<M_TODAY>(out, info); all<M_TODAY_SUB32>(out, info); all<M_PREFIX>(out, info); all<M_PREFIX_PK>(out, info);
    all<M_ONCE_DPP>(out, info); all<M_KEYS>(out, info); all<M_QSAD6>(out, info); all<M_QSAD4>(out, info); all<M_QSAD2>(out, info);
    all<M_ADD64>(out, info);
    return 0;
}
