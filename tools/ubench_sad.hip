// Issue cost on gfx950 of the instructions a byte-cost (SAD / SSD) block matcher can be built
// from, under CONTROLLED occupancy (W = 1 or 2 waves on every SIMD, as tools/ubench_issue.hip),
// plus a semantic check of v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8 against a host model.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sad.hip -o tools/ubench_sad.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32;

enum { OP_BITOP3, OP_QSAD, OP_MQSAD, OP_MQSAD32, OP_SAD, OP_PKSUB, OP_PKMIN, OP_MINU32, OP_MIN3,
       OP_MADU16, OP_ANDOR, OP_LSHLOR, OP_PERM, OP_DOT4, OP_ALIGNBYTE, OP_PKMAD, OP_SUBU32, OP_MSAD,
       OP_QSAD_MIX, OP_BSMIX, OP_COUNT };
static const char *op_name[OP_COUNT] = {
    "v_bitop3_b32", "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8", "v_mqsad_u32_u8", "v_sad_u8", "v_pk_sub_u16",
    "v_pk_min_u16", "v_min_u32", "v_min3_u32", "v_mad_u32_u16", "v_and_or_b32", "v_lshl_or_b32", "v_perm_b32",
    "v_dot4_u32_u8", "v_alignbyte_b32", "v_pk_mad_u16", "v_sub_u32", "v_msad_u8",
    "mix: 6 qsad + 2 pk_sub + 4 and_or/lshl_or + 2 min3", "mix: 15 v_bitop3 + 1 v_alignbit (the bit-sliced kernel's)"};

template <int OP, int W>
__global__ __launch_bounds__(64) void k_rate(u32 *out, u64 *info, int iters)
{
    if (W == 1) asm volatile("" ::: "v250", "a16");
    if (W == 2) asm volatile("" ::: "v200");
    if (W == 3) asm volatile("" ::: "v160");            // 136..168 registers: three waves per SIMD
    if (W == 4) asm volatile("" ::: "v120");            // 104..128: four
    constexpr int ILP = 8;
    u64 a[ILP];
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    v4u q[4];
    const u32 x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u;
    u64 xx = ((u64)x << 32) | (x ^ 0x5bd1e995u);
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = xx + i;
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = v4u{x + i, y, x ^ y, y + i};
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                u32 &lo = reinterpret_cast<u32 *>(&a[i])[0];
                if (OP == OP_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_QSAD) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(xx), "v"(y));
                if (OP == OP_MQSAD) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(xx), "v"(y));
                if (OP == OP_MQSAD32) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(q[i & 3]) : "v"(xx), "v"(y));
                if (OP == OP_SAD) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_MSAD) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_PKSUB) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(lo) : "v"(x));
                if (OP == OP_PKMIN) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(lo) : "v"(x));
                if (OP == OP_MINU32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(lo) : "v"(x));
                if (OP == OP_MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_MADU16) asm volatile("v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(lo) : "v"(y));
                if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_DOT4) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_PKMAD) asm volatile("v_pk_mad_u16 %0, %0, %1, %2 clamp" : "+v"(lo) : "v"(x), "v"(y));
                if (OP == OP_SUBU32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo) : "v"(x));
                if (OP == OP_BSMIX) {
                    if ((r * ILP + i) % 16 == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo) : "v"(x));
                    else asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo) : "v"(x), "v"(y));
                }
            }
            if (OP == OP_QSAD_MIX) {
                // the per-(pixel, 4 shifts) row step of a QSAD block matcher, 2 items side by side
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    u64 t = 0, &A = a[i];
                    u32 &Al = reinterpret_cast<u32 *>(&A)[0], &Ah = reinterpret_cast<u32 *>(&A)[1];
                    u32 &tl = reinterpret_cast<u32 *>(&t)[0], &th = reinterpret_cast<u32 *>(&t)[1];
                    u32 k0, k1, k2, k3, &run = reinterpret_cast<u32 *>(&a[4 + i])[0];
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(t) : "v"(xx), "v"(y));
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(t) : "v"(a[7]), "v"(x));
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(t) : "v"(a[6]), "v"(y));
                    asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Al) : "v"(tl));
                    asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(Ah) : "v"(th));
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(A) : "v"(xx), "v"(x));
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(A) : "v"(a[7]), "v"(y));
                    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(A) : "v"(a[6]), "v"(x));
                    asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(k0) : "v"(Al), "v"(y));
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(k1) : "v"(Al), "v"(x), "v"(y));
                    asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(k2) : "v"(Ah), "v"(y));
                    asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(k3) : "v"(Ah), "v"(x), "v"(y));
                    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(run) : "v"(k0), "v"(k1));
                    asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(run) : "v"(k2), "v"(k3));
                }
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= (u32)a[i] ^ (u32)(a[i] >> 32);
#pragma unroll
    for (int i = 0; i < 4; i++) s ^= q[i].x ^ q[i].y ^ q[i].z ^ q[i].w;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) info[blockIdx.x] = t1 - t0;
}

static int g_iters = 200;
template <int OP, int W>
static void run(u32 *out, u64 *info)
{
    const int iters = g_iters;
    const int grid = 256 * 4 * W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<OP, W>), dim3(grid), dim3(64), 0, 0, out, info, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<u64> h(grid);
    (void)hipMemcpy(h.data(), info, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < grid; b++) cyc += (double)h[b];
    const double n_instr = (double)iters * 16 * (OP == OP_QSAD_MIX ? 28 : 8);
    printf("%-50s %d wave(s)/SIMD: %6.2f shader cycles/instr/wave  -> %5.2f per SIMD   (%.3f ns/instr/SIMD wall, %.2f GHz, %.0f G wave-instr/s chip-wide)\n",
           op_name[OP], W, cyc / grid / n_instr, cyc / grid / n_instr / W, best * 1e6 / (n_instr * W),
           cyc / grid / (best * 1e6), 1024.0 / (best * 1e6 / (n_instr * W)));
}

template <int OP> static void both(u32 *out, u64 *info) { run<OP, 1>(out, info); run<OP, 2>(out, info); }

// ---- semantics -------------------------------------------------------------------------------
__global__ void k_sem(const u64 *s0, const u32 *s1, const u64 *s2, u64 *dq, u64 *dm)
{
    const int i = threadIdx.x;
    dq[i] = __builtin_amdgcn_qsad_pk_u16_u8(s0[i], s1[i], s2[i]);
    dm[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], s2[i]);
}
static u64 model(u64 s0, u32 s1, u64 s2, bool masked)
{
    u64 d = 0;
    for (int i = 0; i < 4; i++) {
        u32 acc = (u32)((s2 >> (16 * i)) & 0xffff);
        for (int j = 0; j < 4; j++) {
            const int a = (int)((s0 >> (8 * (i + j))) & 0xff), b = (int)((s1 >> (8 * j)) & 0xff);
            if (masked && b == 0) continue;
            acc += (u32)abs(a - b);
        }
        d |= (u64)(acc & 0xffff) << (16 * i);
    }
    return d;
}

int main()
{
    u32 *out; u64 *info;
    (void)hipMalloc(&out, 256 * 4 * 4 * 64 * sizeof(u32));
    (void)hipMalloc(&info, 256 * 4 * 4 * sizeof(u64));

    // semantics first
    {
        const int n = 64;
        std::vector<u64> s0(n), s2(n), dq(n), dm(n); std::vector<u32> s1(n);
        srand(7);
        for (int i = 0; i < n; i++) {
            s0[i] = ((u64)rand() << 33) ^ ((u64)rand() << 11) ^ rand();
            s1[i] = ((u32)rand() << 9) ^ rand();
            if (i & 1) s1[i] &= 0xff00ffffu;        // a zero reference byte: the masked form skips it
            if (i % 3 == 0) s1[i] &= 0x000000ffu;
            s2[i] = i < 8 ? 0 : (((u64)rand() << 33) ^ ((u64)rand() << 11) ^ rand());
            if (i == 9) s2[i] = 0xfff0fff0fff0fff0ull;   // wrap-around of the 16-bit accumulators
        }
        u64 *d0, *d2, *d3, *d4; u32 *d1;
        (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 4); (void)hipMalloc(&d2, n * 8);
        (void)hipMalloc(&d3, n * 8); (void)hipMalloc(&d4, n * 8);
        (void)hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(d2, s2.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(n), 0, 0, d0, d1, d2, d3, d4);
        (void)hipMemcpy(dq.data(), d3, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(dm.data(), d4, n * 8, hipMemcpyDeviceToHost);
        int bad_q = 0, bad_m = 0;
        for (int i = 0; i < n; i++) {
            if (dq[i] != model(s0[i], s1[i], s2[i], false)) {
                if (!bad_q) printf("qsad  lane %d: s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", i, s0[i], s1[i], s2[i], dq[i], model(s0[i], s1[i], s2[i], false));
                bad_q++;
            }
            if (dm[i] != model(s0[i], s1[i], s2[i], true)) {
                if (!bad_m) printf("mqsad lane %d: s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", i, s0[i], s1[i], s2[i], dm[i], model(s0[i], s1[i], s2[i], true));
                bad_m++;
            }
        }
        printf("semantics: v_qsad_pk_u16_u8 %d / %d lanes differ from the model (D.u16[i] = S2.u16[i] + sum_j |S0.b[i+j] - S1.b[j]|, wrapping);\n"
               "           v_mqsad_pk_u16_u8 %d / %d (same, reference bytes equal to 0 skipped)\n", bad_q, n, bad_m, n);
    }

    both<OP_BITOP3>(out, info); both<OP_QSAD>(out, info); both<OP_MQSAD>(out, info); both<OP_MQSAD32>(out, info);
    both<OP_SAD>(out, info); both<OP_MSAD>(out, info); both<OP_PKSUB>(out, info); both<OP_PKMIN>(out, info);
    both<OP_MINU32>(out, info); both<OP_MIN3>(out, info); both<OP_MADU16>(out, info); both<OP_ANDOR>(out, info);
    both<OP_LSHLOR>(out, info); both<OP_PERM>(out, info); both<OP_DOT4>(out, info); both<OP_ALIGNBYTE>(out, info);
    both<OP_PKMAD>(out, info); both<OP_SUBU32>(out, info); both<OP_QSAD_MIX>(out, info);
    run<OP_QSAD, 3>(out, info); run<OP_QSAD, 4>(out, info); run<OP_QSAD_MIX, 3>(out, info); run<OP_QSAD_MIX, 4>(out, info);
    run<OP_BITOP3, 3>(out, info); run<OP_BITOP3, 4>(out, info); run<OP_LSHLOR, 3>(out, info); run<OP_LSHLOR, 4>(out, info);
    // the integer-VALU ceiling the chip actually holds: long launches (~100 us, the match kernel's length), 1 .. 4 waves per SIMD
    g_iters = 400;
    printf("-- long launches (%d iterations): what a full-rate stream sustains, by waves per SIMD\n", g_iters);
    run<OP_BITOP3, 1>(out, info); run<OP_BITOP3, 2>(out, info); run<OP_BITOP3, 3>(out, info); run<OP_BITOP3, 4>(out, info);
    run<OP_BSMIX, 1>(out, info); run<OP_BSMIX, 2>(out, info); run<OP_BSMIX, 3>(out, info); run<OP_BSMIX, 4>(out, info);
    return 0;
}
