// sm_cost_qs.hip -- SAD cost mode of the hot path on the quad-SAD unit.
//
// PARITY UNPINNED: the reference has no SAD implementation; the mode is
// the build's own definition (stated at the top of sm_cost.hip; the checker restates it on the CPU).
//
// v_qsad_pk_u16_u8 D, S0 (8 bytes), S1 (4 bytes), S2 (4 x u16) is a block-matching step:
//     D.u16[i] = S2.u16[i] + sum_{j<4} | S0.byte[i+j] - S1.byte[j] |          i = 0..3
// With S1 = four pixels of the left row and S0 = eight pixels of the right row starting at
// x + d, one instruction adds the cost of that 4-pixel group for FOUR consecutive shifts
// d .. d+3 to four packed 16-bit window sums.  It issues at a quarter of the plain VALU rate
// (tools/ubench_sad.hip: 16.8 cycles alone, 12.7 per SIMD with two waves) -- the same
// abs-differences per cycle as v_sad_u8, but the accumulate, the packing and the alignment to
// the shift come for free.
//
// A lane owns PX pixels (4 apart) x 4*NQL shifts and keeps their window sums A as packed u16
// (n*n*255 < 65536: windows up to 15 x 15 -- larger ones, up to the reference's default 21 x 21, keep
// TWO packed sums per shift, see SPLIT below).  Per output row and (pixel, 4 shifts):
//     t = E;   t = qsad(old row groups ..., t)           NG = ceil(n/4) instructions
//     A = qsad(new row groups ..., A)                    NG
//     A -= t                                             2 x v_pk_sub_u16
//     keys (A << 16 | shift), first-wins arg-min         4 + 2 x v_min3_u32
// All 8-byte right-image operands are dword-ALIGNED: a pixel x only takes the shift quads that
// start at d = -((x - half) mod 4) (mod 4), so no byte alignment is ever needed on the right row.
// The left operands (shared by all shifts) are cut with v_alignbyte once per row.
//
// The last group of a window row holds n mod 4 pixels; the other bytes of the LEFT operand are
// zeroed, which makes the instruction add the plain right bytes there: sum_{j >= r} R(p+i+j).
// That term depends on the right row and the position only, so its difference between the row
// that slides in and the row that slides out is computed once per row and right-image position
// by the wave (E, via v_mqsad_pk_u16_u8 against a reference of 255s on exactly those bytes) and
// enters as the INITIAL accumulator of the old row's chain: no VALU instruction is spent on it.
//
// Ghost border: rows / columns outside the image are staged as zeros in both images, which gives
// the oracle's "no tap outside the image, zeros past the right border" -- except for taps LEFT of
// the image (left = 0, right(x' + d) inside).  The columns x < half are therefore recomputed by
// sm_cost_strip.hip, a short launch behind this one (beside it, on a stream of its own, it costs more than it
// takes: sm_cost_wta).
//
// SPLIT (n = 17, 19, 21): a window sum reaches 21 * 21 * 255 = 112 455, more than 16 bits.  The window's
// column groups are split between two packed accumulators -- the first three groups (12 columns: at most
// 12 * 21 * 255 = 64 260) and the rest (at most 9 columns) -- each slid exactly as the single one; the two
// are added as 32-bit integers only where the keys are formed (12 more full-rate instructions per pixel
// and quad, beside 2 x NG quarter-rate v_qsad).  Keys are then sum << 8 | shift (shifts below 256).

#include "sm_internal.h"
#include "sm_cost.h"
#include <type_traits>

typedef unsigned long long u64;
typedef unsigned short v4h __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 pk4_sub(u64 a, u64 b)
{
    return __builtin_bit_cast(u64, (v4h)(__builtin_bit_cast(v4h, a) - __builtin_bit_cast(v4h, b)));
}
// (a & b) | c in one full-rate v_bitop3 (v_and_or_b32 is one of the half-rate class)
__device__ __forceinline__ u32 bop_and_or(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }
__device__ __forceinline__ u64 qsad(u64 r8, u32 l4, u64 acc) { return __builtin_amdgcn_qsad_pk_u16_u8(r8, l4, acc); }

// 8-byte LDS reads at 4-byte alignment (-> ds_read2_b32 into an even register pair)
typedef u64 __attribute__((aligned(4))) u64a4;

template <int N, int NQL, int PX>
__global__ __launch_bounds__(64, 2) void k_sad_qs(const u8 *__restrict__ left, const u8 *__restrict__ right,
                                                  i32 *__restrict__ web, i32 *__restrict__ best,
                                                  const SadGeom g)
{
    constexpr int HALF = N / 2, FG = N / 4, RB = N % 4, NG = FG + 1;
    constexpr u32 MASKR = RB == 1 ? 0x000000ffu : 0x00ffffffu;      // left bytes of the last group
    constexpr u32 MASKC = ~MASKR;                                    // 255 on the bytes zeroed there
    constexpr int WN = NG + PX - 1;                                  // right operands alive per quad
    constexpr bool SPLIT = N * N * 255 >= 65536;                     // two packed sums per shift
    constexpr int GL = SPLIT ? 3 : NG;                               // column groups of the first accumulator
    static_assert(RB == 1 || RB == 3, "odd windows");
    // What must fit 16 bits is each packed sum BETWEEN slides (n rows).  During a slide the entering row is added before
    // the leaving one comes off, so a field holds up to n + 1 rows for an instant (21 x 21: 12 * 22 * 255 = 67 320) and
    // may wrap -- which is exact: v_qsad_pk_u16_u8 and v_pk_sub_u16 work modulo 2^16 PER FIELD, without carry into the
    // neighbouring field and without saturation (tools/ubench_sad.hip checks the wrap against a host model, lane 9),
    // and the difference of the two is below 2^16 again.  A port to a saturating or carry-propagating form must take the
    // leaving row off first.
    static_assert(4 * GL * N * 255 < 65536 && (N - 4 * GL) * N * 255 < 65536, "each packed sum must fit 16 bits between slides");
    static_assert(!SPLIT || NG > GL, "the split needs groups on both sides");
    constexpr int KS = SPLIT ? 8 : 16;                               // bits of the shift in a key
    // "nothing yet": above every real key of a chunk (a real sum is below 2^16 / 2^17) and far enough from 2^32
    // for the chunk bases and the lane's first shift to be added without a wrap
    constexpr u32 KNONE = SPLIT ? 0x7fffff00u : 0xffff0000u;

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

    // ---- stage the tile's rows (+ window halo) with the border rule applied
    smc_stage_rows(lds, L, R, g, xw, ty0, HALF, tid);
    __syncthreads();

    // ---- lane role: residue a, shift-lane sl, pixel group j
    const int a = tid & 3;
    const int sl = (tid >> 2) & (g.nl - 1);
    const int j = tid >> (2 + g.log2nl);
    const int x0 = xw + 4 * PX * j + a;                 // pixel i of this lane: x0 + 4 i
    const int rho = (a - HALF) & 3;                     // (x - HALF) mod 4
    const int bL = (x0 - HALF - rho - (xw - g.padl)) >> 2;      // dword of the window's aligned start
    const int bR = bL + sl * NQL;                       // ... of shift quad 0's right operand
    const int dconst = 4 * sl * NQL - rho;              // shift of (quad 0, position 0)

    u64 A[PX][NQL], A2[SPLIT ? PX : 1][SPLIT ? NQL : 1];            // (A2: the second group of columns, SPLIT only)
#pragma unroll
    for (int i = 0; i < PX; i++)
#pragma unroll
        for (int q = 0; q < NQL; q++) { A[i][q] = 0; if (SPLIT) A2[i][q] = 0; }

    auto ld_pair = [&](const u32 *row, int idx) -> u64 { return *reinterpret_cast<const u64a4 *>(row + idx); };

    // one window row in (rn), one out (ro; none while WARM), optionally the arg-min of row y
    auto step = [&](auto warm_tag, auto out_tag, int rn_i, int ro_i, int y) {
        constexpr bool WARM = decltype(warm_tag)::value, OUT = decltype(out_tag)::value;
        const u32 *rowLn = sL + rn_i * lw, *rowRn = sR + rn_i * rw;
        const u32 *rowLo = sL + ro_i * lw, *rowRo = sR + ro_i * rw;

        // E: per right dword position, (bytes the zeroed left bytes pick up in the new row) - (old row)
#ifndef SAD_EXPERIMENT_NO_E      // (timing experiment only: what the E update and its barrier cost per row)
        for (int k = tid; k < rw - 1; k += 64) {
            const u64 mn = __builtin_amdgcn_mqsad_pk_u16_u8(ld_pair(rowRn, k), MASKC, 0ull);   // 255 (4-RB) - T_new
            u64 e;
            if (WARM) e = pk4_sub(0x0001000100010001ull * (255u * (4 - RB)), mn);
            else e = pk4_sub(__builtin_amdgcn_mqsad_pk_u16_u8(ld_pair(rowRo, k), MASKC, 0ull), mn);
            sE[k] = e;
        }
        __syncthreads();
#endif

        // left operands of this lane's PX windows: NG groups each, 4 pixels apart -> NG + PX - 1 dwords
        u32 un[WN], unp[PX], uo[WN], uop[PX];
        {
            u32 t[WN + 1];
#pragma unroll
            for (int m = 0; m <= WN; m++) t[m] = rowLn[bL + m];
#pragma unroll
            for (int m = 0; m < WN; m++) un[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
#pragma unroll
            for (int i = 0; i < PX; i++) unp[i] = un[i + FG] & MASKR;
            if (!WARM) {
#pragma unroll
                for (int m = 0; m <= WN; m++) t[m] = rowLo[bL + m];
#pragma unroll
                for (int m = 0; m < WN; m++) uo[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
#pragma unroll
                for (int i = 0; i < PX; i++) uop[i] = uo[i + FG] & MASKR;
            }
        }

        // Running minimum in two levels: within a chunk of CH quads the keys carry the shift relative
        // to the chunk (0 .. 4 CH - 1: inline constants -- with the absolute shift in the key the
        // compiler hoists a hundred scalar constants out of the row loop and spills scalar registers
        // into vector ones); the chunks' winners get their bases added at the end of the row.
        // KNONE = "nothing yet" (a real sum is below it, and adding a base cannot wrap).
        constexpr int CH = 16, NCH = (NQL + CH - 1) / CH;
        // (opaque copies: the 2 x NQL uniform comparisons below are otherwise computed once, outside
        // the row loop, and kept in scalar registers -- more than there are)
        int q_last = g.q_last, q_tail = g.q_tail;
        asm volatile("" : "+s"(q_last), "+s"(q_tail));
        int dc = dconst;                    // (the same for the per-lane validity tests of the checked quads)
        asm volatile("" : "+v"(dc));
        u32 runc[NCH][PX];
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int i = 0; i < PX; i++) runc[c][i] = KNONE;

        // operands of quad q: right dwords bR + q + m (m < WN), E of bR + q + FG + i (i < PX).  The reads
        // of quad q + 1 are issued at the top of quad q, and nothing moves across the scheduling
        // barrier between quads: left alone the scheduler hoists the reads of ALL quads to the top
        // of the row and spills hundreds of registers.
        u64 rn[WN + 1], ro[WN + 1], ee[PX + 1];
#pragma unroll
        for (int m = 0; m < WN; m++) {
            rn[m] = ld_pair(rowRn, bR + m);
            if (!WARM) ro[m] = ld_pair(rowRo, bR + m);
        }
#pragma unroll
        for (int i = 0; i < PX; i++) ee[i] = sE[bR + FG + i];
        __builtin_amdgcn_sched_barrier(0);

        // The quads, each nested in the previous one's "q <= q_last" (uniform: beyond q_last no lane
        // has a shift below D), so that leaving early is a jump to the end.  (Not a loop with a break,
        // which the compiler does not unroll -- the sums would live in scratch memory -- and not a
        // condition around each quad's body: the operand windows' rotation, a mere renaming in
        // straight-line code, then becomes 300 register moves per row.)
        auto quad = [&](auto self, auto qtag) __attribute__((always_inline)) -> void {
            constexpr int q = decltype(qtag)::value;
            if constexpr (q < NQL) {
                if (q > q_last) return;
                if (q + 1 < NQL) {
                    rn[WN] = ld_pair(rowRn, bR + q + WN);
                    if (!WARM) ro[WN] = ld_pair(rowRo, bR + q + WN);
                    ee[PX] = sE[bR + q + FG + PX];
                }
#pragma unroll
                for (int i = 0; i < PX; i++) {
                    // (SPLIT: the chains of the first GL groups and of the rest run into accumulators of their
                    // own; E belongs to the LAST group, i.e. to the second chain)
                    u64 t = SPLIT ? 0ull : ee[i], t2 = ee[i];
                    if (!WARM) {
#pragma unroll
                        for (int gp = 0; gp < GL; gp++) t = qsad(ro[i + gp], gp == FG ? uop[i] : uo[i + gp], t);
#pragma unroll
                        for (int gp = GL; gp < NG; gp++) t2 = qsad(ro[i + gp], gp == FG ? uop[i] : uo[i + gp], t2);
                    }
                    u64 acc = A[i][q], acc2 = SPLIT ? A2[i][q] : 0ull;
#pragma unroll
                    for (int gp = 0; gp < GL; gp++) acc = qsad(rn[i + gp], gp == FG ? unp[i] : un[i + gp], acc);
#pragma unroll
                    for (int gp = GL; gp < NG; gp++) acc2 = qsad(rn[i + gp], gp == FG ? unp[i] : un[i + gp], acc2);
                    acc = pk4_sub(acc, t);
                    if (SPLIT) acc2 = pk4_sub(acc2, t2);
                    // (pinned: nothing of a quad may sink below the quads nested in it -- the compiler
                    // otherwise reads all operands on the way in, spilling them, and computes on the way out)
                    asm volatile("" : : "v"(acc));          // (a use only: an output would cost a wait state behind it)
                    A[i][q] = acc;
                    if (SPLIT) { asm volatile("" : : "v"(acc2)); A2[i][q] = acc2; }
                    if (OUT) {
                        // keys: window sum << 16 | shift within the chunk (SPLIT: << 8); the smallest wins,
                        // i.e. the lowest sum and among equals the FIRST shift
                        constexpr int cq = 4 * (q % CH);
                        const u32 lo = (u32)acc, hi = (u32)(acc >> 32);
                        u32 k0, k1, k2, k3;
                        if (SPLIT) {
                            const u32 lo2 = (u32)acc2, hi2 = (u32)(acc2 >> 32);
                            k0 = (((lo & 0xffffu) + (lo2 & 0xffffu)) << KS) | (u32)cq;
                            k1 = (((lo >> 16) + (lo2 >> 16)) << KS) | (u32)(cq + 1);
                            k2 = (((hi & 0xffffu) + (hi2 & 0xffffu)) << KS) | (u32)(cq + 2);
                            k3 = (((hi >> 16) + (hi2 >> 16)) << KS) | (u32)(cq + 3);
                        } else {
                            k0 = (lo << 16) | (u32)cq; k1 = bop_and_or(lo, 0xffff0000u, (u32)(cq + 1));
                            k2 = (hi << 16) | (u32)(cq + 2); k3 = bop_and_or(hi, 0xffff0000u, (u32)(cq + 3));
                        }
                        if (q == 0 || q >= q_tail) {        // uniform: shifts < 0 or >= D may be among these
                            const u32 dlim = (u32)g.D;
                            if ((u32)(dc + 4 * q) >= dlim) k0 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 1) >= dlim) k1 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 2) >= dlim) k2 = 0xffffffffu;
                            if ((u32)(dc + 4 * q + 3) >= dlim) k3 = 0xffffffffu;
                        }
                        u32 r = runc[q / CH][i];
                        r = min(min(r, k0), k1);
                        r = min(min(r, k2), k3);
                        // (pinned: the compiler otherwise sinks the whole min chain to the end of the row
                        // and keeps the keys of every quad alive until then)
                        asm volatile("" : : "v"(r));
                        runc[q / CH][i] = r;
                    }
                }
#pragma unroll
                for (int m = 0; m < WN; m++) { rn[m] = rn[m + 1]; if (!WARM) ro[m] = ro[m + 1]; }
#pragma unroll
                for (int i = 0; i < PX; i++) ee[i] = ee[i + 1];
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, q + 1>{});
            }
        };
        quad(quad, std::integral_constant<int, 0>{});
        u32 run[PX];
        if (OUT) {
#pragma unroll
            for (int i = 0; i < PX; i++) {
                run[i] = runc[0][i];
#pragma unroll
                for (int c = 1; c < NCH; c++) run[i] = min(run[i], runc[c][i] + (u32)(4 * CH * c));
            }
        }

        if (OUT) {
#pragma unroll
            for (int i = 0; i < PX; i++) {
                u32 key = run[i] + (u32)dconst;             // the low half becomes the shift itself (>= 0 for a winner)
                for (int k = 0; k < g.log2nl; k++) key = min(key, (u32)__shfl_xor((int)key, 4 << k));
                const int x = x0 + 4 * i;
                if (sl == 0 && x < g.w) {
                    const size_t o = ((size_t)pair * g.h + y) * g.w + x;
                    web[o] = (i32)(key & ((1u << KS) - 1u)) + 1;
                    if (best) best[o] = (i32)(key >> KS);
                }
            }
        }
        __syncthreads();            // E is rewritten by the next step
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
static const void *sad_qs_ptr(int nql, int px)
{
    if (nql == 33 && px == 2) return (const void *)k_sad_qs<N, 33, 2>;
    if (nql == 17 && px == 2) return (const void *)k_sad_qs<N, 17, 2>;
    if (nql == 17 && px == 4) return (const void *)k_sad_qs<N, 17, 4>;
    if (nql == 9 && px == 4) return (const void *)k_sad_qs<N, 9, 4>;
    if (nql == 5 && px == 4) return (const void *)k_sad_qs<N, 5, 4>;
    return nullptr;
}

// the windows with two packed sums per shift (17, 19, 21): the lane shapes their registers allow
template <int N>
static const void *sad_qs2_ptr(int nql, int px)
{
    if (nql == 17 && px == 2) return (const void *)k_sad_qs<N, 17, 2>;
    if (nql == 9 && px == 2) return (const void *)k_sad_qs<N, 9, 2>;
    if (nql == 5 && px == 4) return (const void *)k_sad_qs<N, 5, 4>;
    return nullptr;
}

// fills g and returns the kernel, or nullptr if this shape is not built (caller falls back)
const void *sm_sad_qs_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *out)
{
    SadGeom g;
    g.rr_stride = 0; g.tbl_pad = 0;         // (the SSD kernels')
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts; g.waves = 1;
    const int half = plan->square_width / 2, n = 2 * half + 1;
    g.ghost = plan->border == SM_GHOST;
    if (n < 3 || n > 21 || g.D > 512 || plan->opt.cost_kernel == 1) return nullptr;     // (512: the entry's own limit)
    const bool split = n > 15;                      // two packed sums per shift: 8 key bits for the shift
    if (split && g.D > 240) return nullptr;
    const int nq = (g.D + 3 + 3) / 4;               // quads that cover shifts -3 .. D-1
    int nql, px;
    if (nq <= 5) { nql = 5; px = 4; }
    else if (nq <= 9) { nql = 9; px = split ? 2 : 4; }
    else if (nq <= 17 || split) { nql = 17; px = split ? 2 : 4; }
    else { nql = 33; px = 2; }
    // (an explicit choice applies where both widths are built; anything else is ignored, not launched)
    if ((plan->opt.cost_pixels_per_lane == 2 || plan->opt.cost_pixels_per_lane == 4) && nql == 17 && !split) px = plan->opt.cost_pixels_per_lane;
    g.nl = 1; g.log2nl = 0;
    while (g.nl * nql < nq) { g.nl <<= 1; g.log2nl++; }
    g.tw = 4 * px * (16 / g.nl);
    g.tiles_x = (g.w + g.tw - 1) / g.tw;
    const int ng = n / 4 + 1;
    g.padl = 4 * ((half + 3 + 3) / 4);
    // left row: dwords bL .. bL + NG + PX - 1 of the last pixel group; right: up to bR + NQL - 1 + NG + PX - 1 (+1 for the pair)
    g.lrow = 8 * ((g.padl + g.tw + 4 * (ng + 1) + 7) / 8);
    g.rrow = 8 * ((g.padl + g.tw + 4 * (g.nl * nql + ng + 2) + 7) / 8);
    // last quad (of the last shift-lane) whose four shifts are all below D for every lane: rho <= 3
    g.q_tail = (g.D - 4 * (g.nl - 1) * nql) / 4;
    if (g.q_tail < 0) g.q_tail = 0;
    // ... and the last quad that holds a shift below D for some lane (rho = 3); with several
    // shift-lanes the lower ones need all their quads
    g.q_last = g.nl > 1 ? nql - 1 : (g.D + 2) / 4;
    if (g.q_last > nql - 1) g.q_last = nql - 1;
    // tile height: whole rounds of two waves per SIMD; rows + warm-up + staging per workgroup
    const int slots = 256 * 4 * 2;
    int best_th = 0; double best_cost = 0;
    for (int th = 8; th <= 128; th += 4) {
        const size_t lds = (size_t)(th + n - 1) * (g.lrow + g.rrow) + 2 * (size_t)g.rrow;
        if (lds > 160 * 1024 / 8) break;
        const long long tiles = (long long)g.tiles_x * ((g.h + th - 1) / th) * pairs;
        const long long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * (th + 0.45 * (n - 1) + 2.0);
        if (!best_th || cost < best_cost) { best_th = th; best_cost = cost; }
    }
    if (!best_th) return nullptr;
    if (plan->opt.cost_tile_h > 0) {         // an explicit tile height, clamped to what a workgroup's LDS holds
        best_th = plan->opt.cost_tile_h;
        while (best_th > 1 && (size_t)(best_th + n - 1) * (g.lrow + g.rrow) + 2 * (size_t)g.rrow > 64 * 1024) best_th--;
    }
    g.tile_h = best_th < g.h ? best_th : g.h;
    g.tiles_y = (g.h + g.tile_h - 1) / g.tile_h;
    g.nsr = g.tile_h + n - 1;
    g.fast_stage = g.w % 4 == 0 && ((uintptr_t)d_left & 3) == 0 && ((uintptr_t)d_right & 3) == 0 &&
                   g.lrow + g.rrow <= 4 * 256;
    g.lds_bytes = g.nsr * (g.lrow + g.rrow) + 2 * g.rrow;
    g.nql = nql; g.px = px;
    const void *fn = nullptr;
    switch (n) {
    case 3: fn = sad_qs_ptr<3>(nql, px); break;
    case 5: fn = sad_qs_ptr<5>(nql, px); break;
    case 7: fn = sad_qs_ptr<7>(nql, px); break;
    case 9: fn = sad_qs_ptr<9>(nql, px); break;
    case 11: fn = sad_qs_ptr<11>(nql, px); break;
    case 13: fn = sad_qs_ptr<13>(nql, px); break;
    case 15: fn = sad_qs_ptr<15>(nql, px); break;
    case 17: fn = sad_qs2_ptr<17>(nql, px); break;
    case 19: fn = sad_qs2_ptr<19>(nql, px); break;
    case 21: fn = sad_qs2_ptr<21>(nql, px); break;
    }
    *out = g;
    return fn;
}
