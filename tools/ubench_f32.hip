// Issue cost on gfx950 of the f32 instructions the edge detector is made of (add, fma, max3, byte
// conversion, select, DPP move) and of their PACKED forms (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32:
// two f32 lanes of a 64-bit register pair per instruction), at 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f32.hip -o tools/ubench_f32.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32;
typedef float f2 __attribute__((ext_vector_type(2)));

enum { OP_ADD, OP_FMA, OP_MAX3, OP_CVT, OP_CNDMASK, OP_DPP, OP_PKADD, OP_PKFMA, OP_PKMUL, OP_PKMOV, OP_MAX, OP_CND_SET, OP_CND_SGPR, OP_CND_NODEP, OP_CMP, OP_CMP_CND, OP_BFI, OP_CND_ADD, OP_CND_E64VCC, OP_CND_SMOV, OP_CND2, OP_EDGEMIX, OP_EDGEMIX_PK, OP_COUNT };
static const char *op_name[OP_COUNT] = {
    "v_add_f32", "v_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_cndmask_b32", "v_mov_b32 dpp wave_shr:1",
    "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_mov_b32", "v_max_f32",
    "v_cndmask_b32 (vcc written by a v_cmp before the loop)", "v_cndmask_b32_e64 (mask in an SGPR pair)",
    "v_cndmask_b32 (vcc; destination is not a source)", "v_cmp_gt_f32 vcc", "v_cmp_gt_f32 vcc + v_cndmask_b32 vcc (pairs)", "v_bfi_b32",
    "v_cndmask_b32 vcc + v_add_f32 alternating (no vcc write)", "v_cndmask_b32_e64 with vcc as the mask pair",
    "s_mov_b64 vcc, s[..] + v_cndmask_b32 vcc (pairs)", "v_cmp_gt_f32 vcc + 2 x v_cndmask_b32 vcc (triples)",
    "edge mix per 2 px, scalar: 8 add + 8 sub + 8 fma(|.|) + 4 max3",
    "edge mix per 2 px, packed: 4 pk_add + 4 pk_add(neg) + 8 pk_fma + 8 max3"};

template <int OP, int W>
__global__ __launch_bounds__(64) void k_rate(float *out, u64 *info, int iters)
{
    if (W == 1) asm volatile("" ::: "v250", "a16");
    if (W == 2) asm volatile("" ::: "v200");
    if (W == 4) asm volatile("" ::: "v120");
    if (W == 8) asm volatile("" ::: "v60");
    constexpr int ILP = 8;
    f2 a[ILP];
    const float x = 0.5f + threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    const f2 xx = {x, y}, yy = {y, x};
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = f2{x + i, y - i};
    u64 smask = 0x5555aaaa0f0ff0f0ull ^ blockIdx.x;
    asm volatile("" : "+s"(smask));
    if (OP == OP_CND_SET || OP == OP_CND_NODEP || OP == OP_CND_ADD || OP == OP_CND_E64VCC) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x), "v"(0.53f) : "vcc");
    float sink[ILP];
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (OP < OP_EDGEMIX) {
#pragma unroll
                for (int i = 0; i < ILP; i++) {
                    float &lo = reinterpret_cast<float *>(&a[i])[0];
                    if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(lo) : "v"(x));
                    if (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(lo) : "v"(x));
                    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, |%2|" : "+v"(lo) : "v"(x), "v"(y));
                    if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y));
                    if (OP == OP_CVT) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(lo));
                    if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(x) : );
                    if (OP == OP_CND_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(x) : );
                    if (OP == OP_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "s"(smask));
                    if (OP == OP_CND_NODEP) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(sink[i]) : "v"(x), "v"(y));
                    if (OP == OP_CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(lo), "v"(x) : "vcc");
                    if (OP == OP_CMP_CND) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(lo) : "v"(x), "v"(y) : "vcc");
                    if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y));
                    if (OP == OP_CND_ADD) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %0, %0, %1" : "+v"(lo) : "v"(x));
                    if (OP == OP_CND_E64VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(lo) : "v"(x));
                    if (OP == OP_CND_SMOV) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(x), "s"(smask) : "vcc");
                    if (OP == OP_CND2) {
                        float &hi = reinterpret_cast<float *>(&a[i])[1];
                        asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
                    }
                    if (OP == OP_DPP) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(lo));
                    if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(xx));
                    if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(xx), "v"(yy));
                    if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(xx));
                    if (OP == OP_PKMOV) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(a[i]) : "v"(xx));
                }
            }
            if (OP == OP_EDGEMIX) {
                // 2 pixels x 4 orientations: s = sa + sb, d = sa - sb, F = fma(s, -T, |d|); max3 x 2 per pixel
                float *f = reinterpret_cast<float *>(a);      // 16 floats: sa/sb of 8 tests
                float F[8];
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    float s, d;
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(f[2 * o]), "v"(f[2 * o + 1]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(f[2 * o]), "v"(f[2 * o + 1]));
                    asm volatile("v_fma_f32 %0, %1, %2, |%3|" : "=v"(F[o]) : "v"(s), "v"(x), "v"(d));
                }
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(F[0]) : "v"(F[0]), "v"(F[1]), "v"(F[2]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[0]) : "v"(F[0]), "v"(F[3]), "v"(f[0]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(F[4]) : "v"(F[4]), "v"(F[5]), "v"(F[6]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[1]) : "v"(F[4]), "v"(F[7]), "v"(f[1]));
            }
            if (OP == OP_EDGEMIX_PK) {
                f2 F1[4], F2[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    f2 s, d;
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(a[2 * o]), "v"(a[2 * o + 1]));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a[2 * o]), "v"(a[2 * o + 1]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(F1[o]) : "v"(s), "v"(xx), "v"(d));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(F2[o]) : "v"(s), "v"(xx), "v"(d));
                }
                float *g1 = reinterpret_cast<float *>(F1), *g2 = reinterpret_cast<float *>(F2);
                float *f = reinterpret_cast<float *>(a);
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    float m;
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(g1[p]), "v"(g1[2 + p]), "v"(g1[4 + p]));
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(g1[6 + p]), "v"(g2[p]));
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(g2[2 + p]), "v"(g2[4 + p]));
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(f[p]) : "v"(m), "v"(g2[6 + p]), "v"(f[p]));
                }
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += a[i].x + a[i].y;
    if (OP == OP_CND_NODEP) {
#pragma unroll
        for (int i = 0; i < ILP; i++) s += sink[i];
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) info[blockIdx.x] = t1 - t0;
}

static int g_iters = 200;
template <int OP, int W>
static void run(float *out, u64 *info)
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
    const double n_instr = (double)iters * 16 * (OP == OP_EDGEMIX ? 28 : OP == OP_EDGEMIX_PK ? 24 : (OP == OP_CMP_CND || OP == OP_CND_ADD || OP == OP_CND_SMOV) ? 16 : OP == OP_CND2 ? 24 : 8);
    printf("%-72s %d wave(s)/SIMD: %6.2f cyc/instr/wave -> %5.2f per SIMD  (%.3f ns/instr/SIMD wall, %.0f G wave-instr/s)%s\n",
           op_name[OP], W, cyc / grid / n_instr, cyc / grid / n_instr / W, best * 1e6 / (n_instr * W),
           1024.0 / (best * 1e6 / (n_instr * W)),
           OP >= OP_EDGEMIX ? "" : "");
    if (OP >= OP_EDGEMIX)
        printf("%-72s    = %.3f ns per pixel-pair per SIMD\n", "", best * 1e6 / ((double)iters * 16 * W));
}

template <int OP> static void all(float *out, u64 *info)
{
    run<OP, 1>(out, info); run<OP, 2>(out, info); run<OP, 4>(out, info); run<OP, 8>(out, info);
}

int main()
{
    float *out; u64 *info;
    (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
    (void)hipMalloc(&info, 256 * 4 * 8 * sizeof(u64));
    if (getenv("UB_CND")) {
        all<OP_CNDMASK>(out, info); all<OP_CND_SET>(out, info); all<OP_CND_SGPR>(out, info); all<OP_CND_NODEP>(out, info);
        all<OP_CMP>(out, info); all<OP_CMP_CND>(out, info); all<OP_BFI>(out, info);
        all<OP_CND_ADD>(out, info); all<OP_CND_E64VCC>(out, info); all<OP_CND_SMOV>(out, info); all<OP_CND2>(out, info);
        return 0;
    }
    all<OP_ADD>(out, info); all<OP_MAX>(out, info); all<OP_FMA>(out, info); all<OP_MAX3>(out, info); all<OP_CVT>(out, info);
    all<OP_CNDMASK>(out, info); all<OP_DPP>(out, info); all<OP_PKADD>(out, info); all<OP_PKFMA>(out, info);
    all<OP_PKMUL>(out, info); all<OP_PKMOV>(out, info); all<OP_EDGEMIX>(out, info); all<OP_EDGEMIX_PK>(out, info);
    return 0;
}
