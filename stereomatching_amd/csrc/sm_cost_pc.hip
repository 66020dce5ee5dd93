// sm_cost_pc.hip -- SAD cost mode of the hot path on the quad-SAD unit, window rows by PREFIX CHAINS.
//
// PARITY UNPINNED: the reference has no SAD implementation; the mode is the build's
// own definition (stated at the top of sm_cost.hip; the checker restates it on the CPU).
//
// Round 5.  k_sad_qs (sm_cost_qs.hip) evaluates every window row from scratch: NG = ceil(n/4) v_qsad for the
// row that slides in and NG for the row that slides out, per (pixel, 4 shifts) -- 6 at 9 x 9, 12 at 21 x 21.
// Here a lane still owns PX pixels 4 apart x 4*NQL shifts with packed u16 window sums W, but the row sums of
// its pixels are formed by SLIDING ALONG THE ROW: with s(x) the n-wide row sum at pixel x and g(c) the cost
// of the four columns c .. c+3,
//         s(x + 4) = s(x) + g(x + h + 1) - g(x - h)                       (h = n/2)
// and what the window sum needs is d(x) = s_new(x) - s_old(x).  v_qsad_pk_u16_u8 adds onto its third
// operand, so two running sums are kept per (lane, 4 shifts) and row, both built from v_qsad alone:
//         P(x + 4) = P(x) + g_new(x + h + 1) + g_old(x - h)          P(x0) = s_new(x0)
//         Q(x + 4) = Q(x) + g_old(x + h + 1) + g_new(x - h)          Q(x0) = s_old(x0)
//         W(x)    += P(x) - Q(x)                                     (2 x v_pk_add_u16 + 2 x v_pk_sub_u16)
// FOUR v_qsad per (pixel, 4 shifts) and row whatever the window (the lane's first pixel: 2 NG, as before),
// all sums modulo 2^16 per field (the window sums themselves fit).  Why not fewer: every scheme that
// evaluates each group once per row (vertical column sums, the review's proposal) has to hand group sums
// from the lane that owns them to the lanes whose windows they enter or leave, and a packed add or a DPP
// move costs a quarter of a v_qsad on this part: tools/ubench_sadmix.hip measures 65.8 ns per step for the
// round-4 mix, 46.6 ns for this one and 43.1 ns for "each group once" with the cheapest exchange (which
// leaves out the seeds and the window's odd column it would need on top).
//
// Alignment.  The leaving group of pixel x starts at x - h, where the pixel's shift quads were aligned in
// the first place (dword-aligned right operands, sm_cost_qs.hip); the entering group starts n = 4 FG + RB
// bytes further on, RB = n mod 4 = 1 or 3.  Its right operands come from a copy of the right row shifted
// by RB bytes, which the wave cuts for the two rows of a step while it prepares E (below); its left
// operands are cut with v_alignbyte once per row like the others.
//
// The lane's first pixel takes its two row sums as k_sad_qs does: the last group of a window row holds RB
// pixels, the left operand's other bytes are zeroed, the right bytes they pick up are E = (new row's) -
// (old row's), computed per right position by the wave and entering as Q's initial value.
//
// Ghost border: as k_sad_qs -- zero staging outside the image, the columns x < half recomputed by
// sm_cost_strip.hip behind this launch.
//
// What else the row loop does NOT pay for (each measured, docs/HISTORY_round5.md section 1):
//   - shifts that do not exist (below 0: a pixel's quads start at -rho; from D on): windows up to 11 x 11 start their sums
//     at a constant no real sum reaches (POISON), so no validity test is left in the loop -- the tests of the checked quads,
//     hoisted by the compiler into every quad, had eaten what the first version saved;
//   - W + P is one 64-bit add where no field can carry (windows up to 9 x 9: ADD64);
//   - the tile's rows are fetched WHILE it is worked on (SmcStream, sm_cost.h): two rows before the first step, row s + 2
//     asked for at step s -- staging all rows first was 5 % of a one-round launch (profiles/r05/ab_sad_knockouts.txt);
//   - workgroups of 1, 2 or 4 waves share the staged rows (the host chooses: at 256 shifts four waves slide 64 rows where a
//     lone wave's 20 KB hold 16).
// PC_EXP: timing knock-outs (results wrong), as MFMA_EXP of sm_cost_mfma.hip.

#include "sm_internal.h"
#include "sm_cost.h"
#include <type_traits>

#ifndef PC_EXP
#define PC_EXP 0        // (timing experiments only, results wrong: 1 rows not loaded from memory, 2 no arg-min keys, 3 no E / shifted rows per step)
#endif

typedef unsigned long long u64;
typedef unsigned short v4h __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ u64 pk4_sub(u64 a, u64 b)
{
    return __builtin_bit_cast(u64, (v4h)(__builtin_bit_cast(v4h, a) - __builtin_bit_cast(v4h, b)));
}
static __device__ __forceinline__ u64 pk4_add(u64 a, u64 b)
{
    return __builtin_bit_cast(u64, (v4h)(__builtin_bit_cast(v4h, a) + __builtin_bit_cast(v4h, b)));
}
// (a & b) | c in one full-rate v_bitop3 (v_and_or_b32 is one of the half-rate class)
static __device__ __forceinline__ u32 bop_and_or(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }
static __device__ __forceinline__ u64 qsad(u64 r8, u32 l4, u64 acc) { return __builtin_amdgcn_qsad_pk_u16_u8(r8, l4, acc); }

// 8-byte LDS reads at 4-byte alignment (-> ds_read2_b32 into an even register pair)
typedef u64 __attribute__((aligned(4))) u64a4;

template <int N, int NQL, int PX>
__global__ __launch_bounds__(256, 2) void k_sad_pc(const u8 *__restrict__ left, const u8 *__restrict__ right,
                                                  i32 *__restrict__ web, i32 *__restrict__ best,
                                                  const SadGeom g)
{
    constexpr int HALF = N / 2, FG = N / 4, RB = N % 4, NG = FG + 1;
    constexpr u32 MASKR = RB == 1 ? 0x000000ffu : 0x00ffffffu;      // left bytes of the last group
    constexpr u32 MASKC = ~MASKR;                                    // 255 on the bytes zeroed there
    constexpr int WNL = NG > PX - 1 ? NG : PX - 1;                   // aligned operands alive per quad
    constexpr int WNS = PX - 1;                                      // shifted ones
    static_assert(RB == 1 || RB == 3, "odd windows");
    static_assert(N * N * 255 < 65536, "a window sum must fit 16 bits (larger windows: k_sad_qs)");
    static_assert(PX >= 2, "a run of pixels per lane");
    constexpr u32 KNONE = 0xffff0000u;
    // Shifts below 0 (the quads start at -rho) and from D on never win.  Windows up to 11 x 11: their sums start
    // at BIG instead of 0 -- a sum is at most SMAX = n n 255 <= 30 855, so BIG + sum stays below 2^16 and above
    // every real sum, and no instruction is spent on them in the row loop.  Larger windows: their keys are
    // replaced where the quad can hold such shifts (the first one and those from q_tail on).
    constexpr u32 SMAX = N * N * 255u;
    constexpr bool POISON = 2 * SMAX + 1 < 65536;
    // Windows up to 9 x 9: W + P as ONE 64-bit add (v_lshl_add_u64) instead of two packed ones -- no field can carry into
    // its neighbour: P <= (n + 3 + 8 (PX - 1)) 255 (its window row, the picked-up bytes, two groups per further pixel),
    // W <= BIG + SMAX.  (Q holds E and may have wrapped: it comes off per field.)
    constexpr u32 PMAX = (N + 3 + 8 * (PX - 1)) * 255u;
    constexpr bool ADD64 = 2 * SMAX + 1 + PMAX < 65536;
    constexpr u32 BIG = ADD64 ? SMAX + 1 : 65535u - SMAX;

    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x;
    const int pair = blockIdx.z;
    const int xw = blockIdx.x * g.tw, ty0 = blockIdx.y * g.tile_h;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *L = left + img, *R = right + img;
    const int lw = g.lrow >> 2, rw = g.rrow >> 2;                    // dwords per staged row
    u32 *sL = lds;                                                   // [nsr][lw]
    u32 *sR = sL + g.nsr * lw;                                       // [nsr][rw]
    u64 *sE = reinterpret_cast<u64 *>(sR + g.nsr * rw);              // [rw]: 4 x u16 per right dword
    u32 *sSn = reinterpret_cast<u32 *>(sE + rw);                     // [rw]: the new row, RB bytes further on
    u32 *sSo = sSn + rw;                                             // [rw]: the old row, likewise

    // ---- stage the tile's rows (+ window halo) with the border rule applied
    // (all waves of the workgroup -- one, two or four, each with its own pixel groups -- stage and share the rows)
    const int nthreads = blockDim.x;
    // Two rows before the first step; row s + 2 is fetched at step s and written to LDS at step s + 1, whose barrier
    // publishes it in time for step s + 2 (smc_stage_rows' slow path -- odd widths, unaligned images -- stages everything here).
    // (the 17-quad builds only: the column pointers cost the smaller ones their third wave per SIMD)
    const bool stream = NQL >= 17 && g.fast_stage != 0;
    SmcStream feed;
    if (stream) feed.setup(L, R, g, xw, tid, nthreads);
#if PC_EXP != 1
    smc_stage_rows(lds, L, R, g, xw, ty0, HALF, tid, 0, nthreads, stream ? 2 : g.nsr);
#else
    for (int k = tid; k < g.nsr * (lw + rw); k += nthreads) lds[k] = (u32)k * 2654435761u;
#endif
    __syncthreads();

    // ---- lane role: residue a, shift-lane sl, pixel group j (over all waves)
    const int a = tid & 3;
    const int sl = (tid >> 2) & (g.nl - 1);
    const int j = tid >> (2 + g.log2nl);
    const int x0 = xw + 4 * PX * j + a;                 // pixel i of this lane: x0 + 4 i
    const int rho = (a - HALF) & 3;                     // (x - HALF) mod 4
    const int bL = (x0 - HALF - rho - (xw - g.padl)) >> 2;      // dword of the window's aligned start
    const int rho2 = (rho + RB) & 3;                    // ... of the groups n bytes further on
    const int bL2 = bL + ((rho + RB) >> 2) + FG;        // dword of pixel 0's entering group
    const int bR = bL + sl * NQL;                       // ... of shift quad 0's right operand
    const int dconst = 4 * sl * NQL - rho;              // shift of (quad 0, position 0)

    u64 W[PX][NQL];
#pragma unroll
    for (int q = 0; q < NQL; q++) {
        u64 w0 = 0;
        if (POISON) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if ((u32)(dconst + 4 * q + e) >= (u32)g.D) w0 |= (u64)BIG << (16 * e);
        }
#pragma unroll
        for (int i = 0; i < PX; i++) W[i][q] = w0;
    }

    auto ld_pair = [&](const u32 *row, int idx) -> u64 { return *reinterpret_cast<const u64a4 *>(row + idx); };

    // one window row in (rn), one out (ro; none while WARM), optionally the arg-min of row y
    auto step = [&](auto warm_tag, auto out_tag, int rn_i, int ro_i, int y) {
        constexpr bool WARM = decltype(warm_tag)::value, OUT = decltype(out_tag)::value;
        const u32 *rowLn = sL + rn_i * lw, *rowRn = sR + rn_i * rw;
        const u32 *rowLo = sL + ro_i * lw, *rowRo = sR + ro_i * rw;

        // the row fetched at the step before goes to LDS, the row two steps ahead is asked for (rn_i is the step's number)
        if (stream && PC_EXP != 1) {
            feed.store(lds);
            if (rn_i + 2 < g.nsr) feed.fetch(g, ty0, HALF, rn_i + 2);
        }
        // per right dword position: E = (bytes the zeroed left bytes pick up in the new row) - (old row), and the
        // two rows RB bytes further on
        for (int k = tid; k < (PC_EXP == 3 ? 0 : rw - 1); k += nthreads) {
            const u32 n0 = rowRn[k], n1 = rowRn[k + 1];
            const u64 mn = __builtin_amdgcn_mqsad_pk_u16_u8(((u64)n1 << 32) | n0, MASKC, 0ull);   // 255 (4-RB) - T_new
            sSn[k] = __builtin_amdgcn_alignbyte(n1, n0, RB);
            u64 e;
            if (WARM) e = pk4_sub(0x0001000100010001ull * (255u * (4 - RB)), mn);
            else {
                const u32 o0 = rowRo[k], o1 = rowRo[k + 1];
                e = pk4_sub(__builtin_amdgcn_mqsad_pk_u16_u8(((u64)o1 << 32) | o0, MASKC, 0ull), mn);
                sSo[k] = __builtin_amdgcn_alignbyte(o1, o0, RB);
            }
            sE[k] = e;
        }
        __syncthreads();

        // left operands: the groups of pixel 0's window (and the leaving groups of the others), 4 bytes apart,
        // and the entering groups n bytes further on
        u32 un[WNL], unp, uen[WNS], uo[WNL], uop = 0, ueo[WNS];
        {
            u32 t[WNL + 1];
#pragma unroll
            for (int m = 0; m <= WNL; m++) t[m] = rowLn[bL + m];
#pragma unroll
            for (int m = 0; m < WNL; m++) un[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
            unp = un[FG] & MASKR;
            u32 t2[WNS + 1];
#pragma unroll
            for (int m = 0; m <= WNS; m++) t2[m] = rowLn[bL2 + m];
#pragma unroll
            for (int m = 0; m < WNS; m++) uen[m] = __builtin_amdgcn_alignbyte(t2[m + 1], t2[m], rho2);
            if (!WARM) {
#pragma unroll
                for (int m = 0; m <= WNL; m++) t[m] = rowLo[bL + m];
#pragma unroll
                for (int m = 0; m < WNL; m++) uo[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
                uop = uo[FG] & MASKR;
#pragma unroll
                for (int m = 0; m <= WNS; m++) t2[m] = rowLo[bL2 + m];
#pragma unroll
                for (int m = 0; m < WNS; m++) ueo[m] = __builtin_amdgcn_alignbyte(t2[m + 1], t2[m], rho2);
            }
        }

        // Running minimum in two levels, as k_sad_qs: keys carry the shift relative to a chunk of CH quads
        // (inline constants), the chunks' winners get their bases added at the end of the row.
        constexpr int CH = 16, NCH = (NQL + CH - 1) / CH;
        int q_last = g.q_last, q_tail = g.q_tail;
        asm volatile("" : "+s"(q_last), "+s"(q_tail));
        int dc = dconst;
        asm volatile("" : "+v"(dc));
        u32 runc[NCH][PX];
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int i = 0; i < PX; i++) runc[c][i] = KNONE;

        // operands of quad q: aligned right dwords bR + q + m (m < WNL), shifted ones bR + q + FG + m (m < WNS),
        // E of bR + q + FG.  The reads of quad q + 1 are issued at the top of quad q, and nothing moves across the
        // scheduling barrier between quads (sm_cost_qs.hip says why).
        u64 rn[WNL + 1], ro[WNL + 1], sn[WNS + 1], so[WNS + 1], ee[2];
#pragma unroll
        for (int m = 0; m < WNL; m++) {
            rn[m] = ld_pair(rowRn, bR + m);
            if (!WARM) ro[m] = ld_pair(rowRo, bR + m);
        }
#pragma unroll
        for (int m = 0; m < WNS; m++) {
            sn[m] = ld_pair(sSn, bR + FG + m);
            if (!WARM) so[m] = ld_pair(sSo, bR + FG + m);
        }
        ee[0] = sE[bR + FG];
        __builtin_amdgcn_sched_barrier(0);

        auto quad = [&](auto self, auto qtag) __attribute__((always_inline)) -> void {
            constexpr int q = decltype(qtag)::value;
            if constexpr (q < NQL) {
                if (q > q_last) return;
                if (q + 1 < NQL) {
                    rn[WNL] = ld_pair(rowRn, bR + q + WNL);
                    sn[WNS] = ld_pair(sSn, bR + q + FG + WNS);
                    if (!WARM) {
                        ro[WNL] = ld_pair(rowRo, bR + q + WNL);
                        so[WNS] = ld_pair(sSo, bR + q + FG + WNS);
                    }
                    ee[1] = sE[bR + q + FG + 1];
                }
                u64 P = 0ull, Q = ee[0];
#pragma unroll
                for (int i = 0; i < PX; i++) {
                    if (i == 0) {
                        // the lane's first pixel: its row sums group by group
#pragma unroll
                        for (int gp = 0; gp < NG; gp++) P = qsad(rn[gp], gp == FG ? unp : un[gp], P);
                        if (!WARM) {
#pragma unroll
                            for (int gp = 0; gp < NG; gp++) Q = qsad(ro[gp], gp == FG ? uop : uo[gp], Q);
                        }
                    } else {
                        // the next pixel of the run: the four columns behind the window in, the window's first four out
                        P = qsad(sn[i - 1], uen[i - 1], P);
                        Q = qsad(rn[i - 1], un[i - 1], Q);
                        if (!WARM) {
                            P = qsad(ro[i - 1], uo[i - 1], P);
                            Q = qsad(so[i - 1], ueo[i - 1], Q);
                        }
                    }
                    const u64 acc = pk4_sub(ADD64 ? W[i][q] + P : pk4_add(W[i][q], P), Q);
                    // (pinned: nothing of a quad may sink below the quads nested in it)
                    asm volatile("" : : "v"(acc));
                    W[i][q] = acc;
                    if (OUT && PC_EXP != 2) {
                        constexpr int cq = 4 * (q % CH);
                        const u32 lo = (u32)acc, hi = (u32)(acc >> 32);
                        u32 k0 = (lo << 16) | (u32)cq, k1 = bop_and_or(lo, 0xffff0000u, (u32)(cq + 1));
                        u32 k2 = (hi << 16) | (u32)(cq + 2), k3 = bop_and_or(hi, 0xffff0000u, (u32)(cq + 3));
                        if (!POISON && (q == 0 || q >= q_tail)) {        // uniform: shifts < 0 or >= D may be among these
                            const u32 dlim = (u32)g.D;
                            if ((u32)(dc + 4 * q) >= dlim) k0 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 1) >= dlim) k1 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 2) >= dlim) k2 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 3) >= dlim) k3 = 0xffffffffu;
                        }
                        u32 r = runc[q / CH][i];
                        r = min(min(r, k0), k1);
                        r = min(min(r, k2), k3);
                        asm volatile("" : : "v"(r));
                        runc[q / CH][i] = r;
                    }
                }
#pragma unroll
                for (int m = 0; m < WNL; m++) { rn[m] = rn[m + 1]; if (!WARM) ro[m] = ro[m + 1]; }
#pragma unroll
                for (int m = 0; m < WNS; m++) { sn[m] = sn[m + 1]; if (!WARM) so[m] = so[m + 1]; }
                ee[0] = ee[1];
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, q + 1>{});
            }
        };
        quad(quad, std::integral_constant<int, 0>{});

        if (OUT) {
            u32 key[PX];
#pragma unroll
            for (int i = 0; i < PX; i++) {
                u32 run = runc[0][i];
#pragma unroll
                for (int c = 1; c < NCH; c++) run = min(run, runc[c][i] + (u32)(4 * CH * c));
                key[i] = run + (u32)dconst;                 // the low half becomes the shift itself (>= 0 for a winner)
            }
            // the shift lanes of a pixel group: all PX exchanges of a level in flight together (one wait per level, not per pixel)
            for (int k = 0; k < g.log2nl; k++) {
                u32 other[PX];
#pragma unroll
                for (int i = 0; i < PX; i++) other[i] = (u32)__shfl_xor((int)key[i], 4 << k);
#pragma unroll
                for (int i = 0; i < PX; i++) key[i] = min(key[i], other[i]);
            }
            if (sl == 0) {
                i32 *wrow = web + ((size_t)pair * g.h + y) * g.w + x0;
                i32 *brow = best ? best + ((size_t)pair * g.h + y) * g.w + x0 : nullptr;
#pragma unroll
                for (int i = 0; i < PX; i++) {
                    if (x0 + 4 * i < g.w) {
                        wrow[4 * i] = (i32)(key[i] & 0xffffu) + 1;
                        if (brow) brow[4 * i] = (i32)(key[i] >> 16);
                    }
                }
            }
        }
        __syncthreads();            // E and the shifted rows are rewritten by the next step
    };

    const int rows_out = min(g.tile_h, g.h - ty0);
    using T = std::true_type;
    using F = std::false_type;
    // staged row e is image row ty0 - HALF + e: output row t has window rows t .. t + N - 1
#pragma unroll 1
    for (int e = 0; e < N - 1; e++) step(T{}, F{}, e, 0, 0);
    step(T{}, T{}, N - 1, 0, ty0);
#pragma unroll 1
    for (int t = 1; t < rows_out; t++) step(F{}, T{}, t + N - 1, t - 1, ty0 + t);
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------

template <int N>
static const void *sad_pc_ptr(int nql)
{
    if (nql == 17) return (const void *)k_sad_pc<N, 17, 4>;
    if (nql == 9) return (const void *)k_sad_pc<N, 9, 4>;
    if (nql == 5) return (const void *)k_sad_pc<N, 5, 4>;
    return nullptr;
}

// fills g and returns the kernel, or nullptr if this shape is not built (caller falls back to k_sad_qs)
const void *sm_sad_pc_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *out)
{
    SadGeom g;
    g.rr_stride = 0; g.tbl_pad = 0;         // (the SSD kernels')
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts;
    const int half = plan->square_width / 2, n = 2 * half + 1;
    g.ghost = plan->border == SM_GHOST;
    if (n < 3 || n > 15 || g.D > 512 || plan->opt.cost_kernel == 1 || plan->opt.cost_kernel == 4) return nullptr;
    const int nq = (g.D + 3 + 3) / 4;               // quads that cover shifts -3 .. D-1
    const int px = 4;
    const int nql = nq <= 5 ? 5 : nq <= 9 ? 9 : 17;
    g.nl = 1; g.log2nl = 0;
    while (g.nl * nql < nq) { g.nl <<= 1; g.log2nl++; }
    if (g.nl > 16) return nullptr;
    const int ng = n / 4 + 1;
    g.padl = 4 * ((half + 3 + 3) / 4);
    g.q_tail = (g.D - 4 * (g.nl - 1) * nql) / 4;
    if (g.q_tail < 0) g.q_tail = 0;
    g.q_last = g.nl > 1 ? nql - 1 : (g.D + 2) / 4;
    if (g.q_last > nql - 1) g.q_last = nql - 1;
    // Workgroup = 1, 2 or 4 waves side by side (each its own 64 / nl pixel groups) sharing the staged rows: with many
    // shifts a lone wave's tile is narrow (64 pixels at 256 shifts) under a right-image span of 4 (nl nql + ..) bytes,
    // its 20 KB of LDS hold few rows and the n - 1 warm-up rows weigh a quarter of the launch (C5: 16-row tiles);
    // four waves share one span and slide 64 rows.  Tile height and workgroup width together: whole rounds of two
    // waves per SIMD, rows + warm-up (a warm-up row costs ~0.41 of an output row here) + staging per workgroup.
    const int slots = 256 * 4 * 2;
    int best_th = 0, best_wv = 0; double best_cost = 0;
    auto row_bytes = [&](int wv, int *lrow, int *rrow) {
        const int tw = 4 * px * (16 / g.nl) * wv;
        // left row: dwords up to bL2 + PX - 1 (<= bL + FG + PX) of the last pixel group; right: aligned up to
        // bR + NQL - 1 + max(NG, PX - 1) + 1, shifted up to bR + NQL - 1 + FG + PX - 1 + 1 (+1: the bytes they are cut from)
        *lrow = 8 * ((g.padl + tw + 4 * (ng + 3) + 7) / 8);
        *rrow = 8 * ((g.padl + tw + 4 * (g.nl * nql + ng + px + 2) + 7) / 8);
        return tw;
    };
    for (int wv = 1; wv <= 4; wv *= 2) {
        int lrow, rrow;
        const int tw = row_bytes(wv, &lrow, &rrow);
        if (wv > 1 && tw / 2 >= g.w) break;              // (a workgroup wider than the image)
        if (lrow + rrow > 4 * 4 * 64 * wv) continue;     // (the fast staging path's reach)
        const int tiles_x = (g.w + tw - 1) / tw;
        for (int th = 8; th <= 128; th += 4) {
            const size_t lds = (size_t)(th + n - 1) * (lrow + rrow) + 4 * (size_t)rrow;
            if (lds > (size_t)wv * 160 * 1024 / 8) break;
            const long long waves = (long long)tiles_x * ((g.h + th - 1) / th) * pairs * wv;
            const long long rounds = (waves + slots - 1) / slots;
            const double cost = (double)rounds * (th + 0.41 * (n - 1) + 2.0);
            if (!best_th || cost < best_cost * (wv > best_wv ? 0.97 : 1.0)) { best_th = th; best_wv = wv; best_cost = cost; }
        }
    }
    if (!best_th) return nullptr;
    int forced_wv = 0;
    if (plan->opt.cost_workgroup_waves == 1 || plan->opt.cost_workgroup_waves == 2 || plan->opt.cost_workgroup_waves == 4) {
        int lrow, rrow;                       // an explicit width applies where the staging path reaches it
        row_bytes(plan->opt.cost_workgroup_waves, &lrow, &rrow);
        if (lrow + rrow <= 4 * 4 * 64 * plan->opt.cost_workgroup_waves) forced_wv = plan->opt.cost_workgroup_waves;
    }
    if (forced_wv && forced_wv != best_wv) {
        // (the height the model gives that width)
        int lrow, rrow;
        const int tw = row_bytes(forced_wv, &lrow, &rrow);
        const int tiles_x = (g.w + tw - 1) / tw;
        best_th = 0;
        for (int th = 8; th <= 128; th += 4) {
            const size_t lds = (size_t)(th + n - 1) * (lrow + rrow) + 4 * (size_t)rrow;
            if (lds > (size_t)forced_wv * 160 * 1024 / 8) break;
            const long long waves = (long long)tiles_x * ((g.h + th - 1) / th) * pairs * forced_wv;
            const long long rounds = (waves + slots - 1) / slots;
            const double cost = (double)rounds * (th + 0.41 * (n - 1) + 2.0);
            if (!best_th || cost < best_cost) { best_th = th; best_cost = cost; }
        }
        if (!best_th) best_th = 8;
        best_wv = forced_wv;
    }
    g.waves = best_wv;
    g.tw = row_bytes(g.waves, &g.lrow, &g.rrow);
    g.tiles_x = (g.w + g.tw - 1) / g.tw;
    const size_t lds_cap = (size_t)g.waves * 160 * 1024 / 8;      // (beyond 64 KB: sm_cost_wta raises the kernel's limit)
    if (plan->opt.cost_tile_h > 0) best_th = plan->opt.cost_tile_h;      // an explicit tile height, clamped to what a workgroup's LDS holds
    while (best_th > 1 && (size_t)(best_th + n - 1) * (g.lrow + g.rrow) + 4 * (size_t)g.rrow > lds_cap) best_th--;
    g.tile_h = best_th < g.h ? best_th : g.h;
    g.tiles_y = (g.h + g.tile_h - 1) / g.tile_h;
    g.nsr = g.tile_h + n - 1;
    g.fast_stage = g.w % 4 == 0 && ((uintptr_t)d_left & 3) == 0 && ((uintptr_t)d_right & 3) == 0 &&
                   g.lrow + g.rrow <= 4 * 4 * 64 * g.waves;
    g.lds_bytes = g.nsr * (g.lrow + g.rrow) + 4 * g.rrow;
    g.nql = nql; g.px = px;
    const void *fn = nullptr;
    switch (n) {
    case 3: fn = sad_pc_ptr<3>(nql); break;
    case 5: fn = sad_pc_ptr<5>(nql); break;
    case 7: fn = sad_pc_ptr<7>(nql); break;
    case 9: fn = sad_pc_ptr<9>(nql); break;
    case 11: fn = sad_pc_ptr<11>(nql); break;
    case 13: fn = sad_pc_ptr<13>(nql); break;
    case 15: fn = sad_pc_ptr<15>(nql); break;
    }
    *out = g;
    return fn;
}
