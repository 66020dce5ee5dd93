// sm_cost_mfma.hip -- SSD cost mode of the hot path on the matrix cores (v_mfma_i32_32x32x32_i8).
//
// PARITY UNPINNED: the reference has no SSD implementation; the mode is the
// build's own definition (top of sm_cost.hip; the checker restates it on the CPU).
//
//   SSD_d(x, y) = LL(x, y) + RR(x + d, y) - 2 LR_d(x, y)       (window sums of squares / of products)
// The only per-shift work is LR, and with u = x + d (the RIGHT image position) it is a plain matrix product:
//     LR[u][x] = sum over window rows, sum over t < n of  R[row][u + t - half] * L[row][x + t - half]
// i.e. C = A B with A[u][k] = R[u + k - half] (M = right positions), B[k][x] = L[x + k - half] (N = pixels)
// and K = the n columns of a window row.  Only the band 0 <= u - x < D is wanted: for a tile of 32 pixels,
// (D + 31) / 32 blocks of 32 right positions, of which the first and the last are half used (D = 256: 89 %
// of the products are wanted ones).
//
// One wave per workgroup, a tile of 32 pixels x tile_h rows, NB accumulator blocks of 32 x 32 int32 in
// VGPRs that SLIDE down the image: per output row ONE v_mfma_i32_32x32x32_i8 per block adds the row that
// enters the window (K-slots 0..15: n bytes of it, the rest zero in B) and takes the row that leaves it
// off (K-slots 16..31).  Pixels are staged as SIGNED bytes, pixel - 128 (differences do not care), and
// the leaving row's LEFT operand is complemented: ~l = -l - 1, so its products arrive negated plus
// -sum R[u + t - half] -- a drift that depends on u only, and is folded into the RR table where that row's
// squares come off anyway.  What a K-slot's index k is inside the instruction does not matter to a product
// as long as A and B agree (tools/ubench_mfma_i8.hip checks the layout used here against a host product).
//
// A lane holds ONE pixel (x = lane % 32) and 16 right positions per block: the arg-min over the shifts is
// in-register, 1.5 instructions per (pixel, shift) -- v_lshl_add_u32 forms -key = (LR << 9) + entry(u),
// entry(u) = -(RR(u) << 8) - (u & 255) read as one ds_read_b128 per four positions, and one v_max3_i32 takes
// two keys -- then one exchange between the lane halves.  Measured at C5 (4K, 256 shifts, 11 x 11): 4.06
// VALU lane-instructions per (pixel, shift), RR table and operand set-up included, against 14.5 for the
// dot-product kernel (sm_cost_ssd.hip); 403 us per launch against 974; the matrix pipe is busy 11 % of
// the time (profiles/r04/cost_ssd_C5.json).
//
// Ghost border: as in the other two kernels, rows / columns outside the image are staged as zero pixels in
// both images; the columns x < half are recomputed by the masked kernel of sm_cost.hip.
//
// Limits: windows up to 11 x 11 (the key holds RR - 2 LR in 24 signed bits), D <= 256.

#include "sm_internal.h"
#include "sm_cost.h"
#include <type_traits>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ u32 sdot4(u32 a, u32 b, u32 acc) { return (u32)__builtin_amdgcn_sdot4((int)a, (int)b, (int)acc, false); }

#ifndef MFMA_EXP
#define MFMA_EXP 0      // (timing experiments only, results wrong: 1 no table reads, 2 no table update, 3 no MFMA, 4 no keys)
#endif
// waves per SIMD the register count allows: the NB x 16 accumulators are most of it
#ifndef MFMA_W3
#define MFMA_W3 5
#endif
#ifndef MFMA_W4
#define MFMA_W4 3
#endif
constexpr int mfma_waves(int nb) { return nb <= MFMA_W4 ? 4 : nb <= MFMA_W3 ? 3 : 2; }

template <int N, int NB>
__global__ __launch_bounds__(256, mfma_waves(NB)) void k_ssd_mfma(const u8 *__restrict__ left, const u8 *__restrict__ right,
                                                    i32 *__restrict__ web, i32 *__restrict__ best,
                                                    const SadGeom g)
{
    constexpr int HALF = N / 2, FD = N / 4, RB = N % 4;             // full dwords of a window row, bytes of the last
    constexpr u32 MASKR = RB == 1 ? 0x000000ffu : 0x00ffffffu;
    constexpr i32 NONE = (i32)0x80000100;                            // -(0x7fffff00): loses to every key
    static_assert(N >= 3 && N <= 11 && (RB == 1 || RB == 3), "odd windows up to 11 x 11");
    static_assert(N * N * 65025 < (1 << 23), "RR - 2 LR must fit 24 signed bits of the key");
    static_assert(NB >= 1 && NB <= 9, "D <= 256");

    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    // Round 5: g.waves (1, 2 or 4) waves per workgroup, side by side, 32 pixels each, SHARE the staged rows: one right-image
    // span of 32 (NB + waves - 1) positions instead of `waves` spans of 32 NB -- the LDS a wave needs per row falls (C5: 384 ->
    // 224 bytes at two waves), the tile gets taller, the n - 1 warm-up rows weigh less, and the launch fetches its inputs
    // half as often.  Every wave keeps a table of its own and never waits for its neighbours after the staging.
    const int tid = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.z;
    const int xw = blockIdx.x * g.tw, ty0 = blockIdx.y * g.tile_h;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *L = left + img, *R = right + img;
    const int lw = g.lrow >> 2, rw = g.rrow >> 2;
    u32 *sL = lds;                                                   // [nsr][lw]
    u32 *sR = sL + g.nsr * lw;                                       // [nsr][rw]
    // entry(u) = -(T(u) << 8) - (u & 255), T = RR minus twice the drift of the LR sums (see the step); 16-byte aligned
    u32 *sT = sR + g.nsr * rw + g.tbl_pad + wave * (32 * NB + 4);

    smc_stage_rows(lds, L, R, g, xw, ty0, HALF, (int)threadIdx.x, 0x80808080u, (int)blockDim.x);   // signed bytes: pixel - 128
    for (int u = tid; u < 32 * NB; u += 64) sT[u] = (u32)(-(u & 255));
    __syncthreads();

    const int xl = tid & 31, h = tid >> 5;
    // byte padl + xl - HALF of a staged row: the window start of pixel xl, and of right position u = xl (block 0)
    const int ob = g.padl + 32 * wave + xl - HALF;
    const int bw = ob >> 2;
    const u32 rho = (u32)(ob & 3);
    const u32 cm = h ? 0xffffffffu : 0u;                // the leaving row's left operand is complemented
    const u32 c2 = h ? 0x02020202u : 0u;                // ... and its right operand's squares carry 2 x the row sum
    const int sgn = h ? 256 : -256;                     // entry -= new << 8 (lanes h = 0), += old << 8 (h = 1)
    const int lo = xl - 4 * h;                          // block 0: position 8 q + 4 h + j is a shift >= 0 iff 8 q + j >= lo
    const int hi = g.D + xl - 4 * h;                    // shift < D iff 32 b + 8 q + j < hi

    bool tri[16];                                       // lane masks, once per kernel: position 8 q + 4 h + j >= the pixel's
#pragma unroll
    for (int r = 0; r < 16; r++) tri[r] = 8 * (r / 4) + r % 4 >= lo;
    const bool d32 = (g.D & 31) == 0;

    v16i acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[b][r] = 0;
    i32 LLs = 0;

    // One pass over the NB blocks does two rows' work, block by block: the arg-min of the row the accumulators
    // hold (KEYS: row yk), then the next window row in (FEED: staged row rn_i enters, ro_i leaves unless WARM).
    // Block b's keys read its accumulators and table entries just before block b's MFMA and table update
    // change them, so the matrix pipe works on row y + 1 while the VALU ranks row y, and a block's result is
    // not asked for until a whole pass later.
    auto step = [&](auto keys_tag, auto feed_tag, auto warm_tag, int rn_i, int ro_i, int yk) {
        constexpr bool KEYS = decltype(keys_tag)::value, FEED = decltype(feed_tag)::value, WARM = decltype(warm_tag)::value;
        const int rsel = h ? ro_i : rn_i;               // lanes 0..31 feed the entering row, 32..63 the leaving one
        const u32 *rowL = sL + rsel * lw + bw, *rowR = sR + rsel * rw + bw;
        const bool live = !WARM || h == 0;              // (while WARM nothing leaves: the upper half feeds zeros)
        const int sg = live ? sgn : 0;
        const i32 llk = LLs;                            // LL of row yk (this lane half's share)

        // B operand: 16 bytes of the left row from the pixel's window start, bytes >= N zero
        v4i bop = {0, 0, 0, 0};
        if (FEED) {
            u32 t[5], v[4];
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = rowL[k];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = __builtin_amdgcn_alignbyte(t[k + 1], t[k], rho);
            if (best) {                                 // uniform: LL of the pixel's window row (before complementing)
                u32 s = 0;
#pragma unroll
                for (int k = 0; k <= FD; k++) { const u32 q = k == FD ? v[k] & MASKR : v[k]; s = sdot4(q, q, s); }
                if (live) LLs += h ? -(i32)s : (i32)s;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 q = v[k] ^ cm;
                if (k == FD) q &= MASKR;
                if (k > FD || !live) q = 0;
                bop[k] = (int)q;
            }
        }

        // (the right row's dwords are requested PF blocks ahead, the table entries one block ahead: between the
        // scheduling barriers that keep the register count down nothing else hides the LDS latency)
        constexpr int PF = 2;
        u32 tq[PF + 1][5];
        v4i eq[2][4];
        if (FEED) {
#pragma unroll
            for (int b = 0; b < PF && b < NB; b++)
#pragma unroll
                for (int k = 0; k < 5; k++) tq[b][k] = rowR[8 * b + k];
        }
        if (KEYS) {
#pragma unroll
            for (int q = 0; q < 4; q++) eq[0][q] = *reinterpret_cast<const v4i *>(sT + 8 * q + 4 * h);
        }
        i32 run0 = NONE, run1 = NONE;                   // positions 0..255 (blocks 0..7) and 256.. (block 8)
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (FEED && b + PF < NB) {
#pragma unroll
                for (int k = 0; k < 5; k++) tq[(b + PF) % (PF + 1)][k] = rowR[8 * (b + PF) + k];
            }
#if MFMA_EXP != 1
            if (KEYS && b + 1 < NB) {
#pragma unroll
                for (int q = 0; q < 4; q++) eq[(b + 1) & 1][q] = *reinterpret_cast<const v4i *>(sT + 32 * (b + 1) + 8 * q + 4 * h);
            }
#endif
            if (KEYS && MFMA_EXP != 4) {
                i32 keys[16];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int j = 0; j < 4; j++) keys[4 * q + j] = (i32)(((u32)acc[b][4 * q + j] << 9) + (u32)eq[b & 1][q][j]);
                // The band: block 0 holds shifts < 0 (position 8 q + 4 h + j left of the pixel, 8 q + j < lo), the last
                // block -- and the one before it unless D is a multiple of 32 -- shifts >= D (32 b + 8 q + j >= hi).
                // For D a multiple of 32 the last block's bad positions are exactly block 0's good ones.
                if (b == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) keys[r] = tri[r] ? keys[r] : NONE;
                }
                if (b >= NB - 2) {
                    if (d32) {                           // uniform
                        if (b == NB - 1) {
#pragma unroll
                            for (int r = 0; r < 16; r++) keys[r] = tri[r] ? NONE : keys[r];
                        }
                    } else {
                        int hv = hi;
                        asm volatile("" : "+v"(hv));     // (not hoisted out of the row loop: 32 lane masks would be)
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            if (32 * b + 8 * (r / 4) + r % 4 >= hv) keys[r] = NONE;
                    }
                }
                i32 &run = b < 8 ? run0 : run1;
#pragma unroll
                for (int r = 0; r < 16; r += 2) run = max(max(run, keys[r]), keys[r + 1]);
                asm volatile("" : "+v"(run));            // (the maxima are otherwise deferred to the row's end, every key alive)
            }
            if (FEED) {
                // A operand: 16 bytes of the right row from position 32 b + xl's window start
                const u32 *t = tq[b % (PF + 1)];
                v4i aop;
#pragma unroll
                for (int k = 0; k < 4; k++) aop[k] = (int)__builtin_amdgcn_alignbyte(t[k + 1], t[k], rho);
#if MFMA_EXP != 3
                acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop, bop, acc[b], 0, 0, 0);
#else
                acc[b][0] += aop[0] ^ bop[0]; acc[b][5] += aop[1] ^ bop[1]; acc[b][10] += aop[2] ^ bop[2]; acc[b][15] += aop[3] ^ bop[3];
#endif
                // T(u) += the entering row's squares; -= the leaving row's squares + 2 x its sum (the drift of the LR sums:
                // the complemented left operand leaves -sum R behind in every accumulator of position u)
#if MFMA_EXP != 2
                u32 s = 0;
#pragma unroll
                for (int k = 0; k <= FD; k++) {
                    const u32 q = k == FD ? (u32)aop[k] & MASKR : (u32)aop[k];
                    s = sdot4(q, q, s);
                    s = sdot4(q, c2, s);
                }
                __hip_atomic_fetch_add(&sT[32 * b + xl], (u32)__mul24((int)s, sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);          // (block by block: the scheduler otherwise keeps every block's operands alive)
        }

        if (KEYS) {
            // back to keys: (RR - 2 LR) << 8 | position & 255; the first position wins among equals
            i32 k0 = -run0, v = k0 >> 8, d = (k0 & 255) - xl;
            if (NB > 8) {
                const i32 k1 = -run1, v1 = k1 >> 8;
                if (v1 < v) { v = v1; d = 256 + (k1 & 255) - xl; }
            }
            // the other half of the wave holds the other positions of the same pixel
            const i32 vo = __shfl_xor(v, 32), dd = __shfl_xor(d, 32);
            if (vo < v || (vo == v && dd < d)) { v = vo; d = dd; }
            const i32 ll = best ? llk + __shfl_xor(llk, 32) : 0;
            const int x = xw + 32 * wave + xl;
            if (h == 0 && x < g.w && !(g.ghost && x < HALF)) {
                const size_t o = ((size_t)pair * g.h + yk) * g.w + x;
                web[o] = d + 1;
                if (best) best[o] = v + ll;
            }
        }
    };

    const int rows_out = min(g.tile_h, g.h - ty0);
    using T = std::true_type;
    using F = std::false_type;
    // staged row e is image row ty0 - HALF + e: output row t has window rows t .. t + N - 1
#pragma unroll 1
    for (int e = 0; e < N; e++) step(F{}, T{}, T{}, e, 0, 0);
#pragma unroll 1
    for (int t = 1; t < rows_out; t++) step(T{}, T{}, F{}, t + N - 1, t - 1, ty0 + t - 1);
    step(T{}, F{}, F{}, 0, 0, ty0 + rows_out - 1);
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------

template <int N>
static const void *mfma_ptr(int nb)
{
    switch (nb) {
    case 1: return (const void *)k_ssd_mfma<N, 1>;
    case 2: return (const void *)k_ssd_mfma<N, 2>;
    case 3: return (const void *)k_ssd_mfma<N, 3>;
    case 4: return (const void *)k_ssd_mfma<N, 4>;
    case 5: return (const void *)k_ssd_mfma<N, 5>;
    case 6: return (const void *)k_ssd_mfma<N, 6>;
    case 7: return (const void *)k_ssd_mfma<N, 7>;
    case 8: return (const void *)k_ssd_mfma<N, 8>;
    case 9: return (const void *)k_ssd_mfma<N, 9>;
    }
    return nullptr;
}

// fills g and returns the kernel, or nullptr if this shape is not built (caller falls back)
const void *sm_ssd_mfma_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *out)
{
    SadGeom g;
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts; g.waves = 1;
    const int half = plan->square_width / 2, n = 2 * half + 1;
    g.ghost = plan->border == SM_GHOST;
    if (n < 3 || n > 11 || g.D > 256 || plan->opt.cost_kernel == 1 || plan->opt.cost_kernel == 2) return nullptr;
    const int nb = (g.D + 31 + 31) / 32;                // right positions 0 .. D + 30
    g.nl = 1; g.log2nl = 0; g.nql = 0; g.px = 1; g.q_tail = 0; g.q_last = 0;
    g.padl = 4 * ((half + 3 + 3) / 4);
    const int per_simd = mfma_waves(nb);
    const int slots = 256 * 4 * per_simd;
    // workgroup width (1, 2 or 4 waves sharing the staged rows) and tile height together: whole rounds of the waves
    // the registers allow per SIMD; rows + warm-up (a warm-up row costs ~0.6 of an output row) + staging per wave
    auto shape = [&](int wv, int *lrow, int *rrow, int *tbl) {
        // a lane reads 5 dwords from dword (padl + 32 wave + xl - half) / 4 (+ 8 b in the right row)
        *lrow = 8 * ((g.padl + 32 * wv + 24 + 7) / 8);
        *rrow = 8 * ((g.padl + 32 * (nb + wv - 1) + 24 + 7) / 8);
        *tbl = wv * (4 * 32 * nb + 16);      // the tables: 32 nb entries per wave, 16-byte aligned behind the staged rows
    };
    // (the plan's own choice is between ONE and FOUR waves: two were measured no better than one at equal tile heights and
    // slower at the taller tiles the model gives them -- C5: 0.427 / 0.447 / 0.391 ms at 1 / 2 / 4 waves,
    // profiles/r05/ab_ssd_workgroup_waves.txt -- for a reason that was not found; an explicit 2 is honoured)
    int best_th = 0, best_wv = 1; double best_cost = 0;
    for (int wv = 1; wv <= 4; wv *= 2) {
        if (plan->opt.cost_workgroup_waves ? plan->opt.cost_workgroup_waves != wv : wv == 2) continue;
        int lrow, rrow, tbl;
        shape(wv, &lrow, &rrow, &tbl);
        if (wv > 1 && 32 * wv / 2 >= g.w) break;
        if (lrow + rrow > 4 * 4 * 64 * wv) continue;         // (the fast staging path's reach)
        const int tiles_x = (g.w + 32 * wv - 1) / (32 * wv);
        for (int th = 8; th <= 128; th += 4) {
            const size_t lds = (size_t)(th + n - 1) * (lrow + rrow) + tbl;
            if (lds > (size_t)wv * (160 * 1024 / (4 * per_simd)) || lds > 64 * 1024) break;
            const long long waves = (long long)tiles_x * ((g.h + th - 1) / th) * pairs * wv;
            const long long rounds = (waves + slots - 1) / slots;
            const double cost = (double)rounds * (th + 0.6 * (n - 1) + 2.0);
            if (!best_th || cost < best_cost * (wv > best_wv ? 0.97 : 1.0)) { best_th = th; best_wv = wv; best_cost = cost; }
        }
    }
    if (!best_th) return nullptr;
    g.waves = best_wv;
    int tbl_bytes;
    shape(g.waves, &g.lrow, &g.rrow, &tbl_bytes);
    g.tw = 32 * g.waves;
    g.tiles_x = (g.w + g.tw - 1) / g.tw;
    if (plan->opt.cost_tile_h > 0) {         // an explicit tile height, clamped to what a workgroup's LDS holds
        best_th = plan->opt.cost_tile_h;
        while (best_th > 1 && (size_t)(best_th + n - 1) * (g.lrow + g.rrow) + tbl_bytes > 64 * 1024) best_th--;
    }
    g.tile_h = best_th < g.h ? best_th : g.h;
    g.tiles_y = (g.h + g.tile_h - 1) / g.tile_h;
    g.nsr = g.tile_h + n - 1;
    g.fast_stage = g.w % 4 == 0 && ((uintptr_t)d_left & 3) == 0 && ((uintptr_t)d_right & 3) == 0 &&
                   g.lrow + g.rrow <= 4 * 4 * 64 * g.waves;
    // dwords between the end of the staged rows and the table: whatever makes the table 16-byte aligned
    g.rr_stride = 0;
    g.tbl_pad = (4 - (g.nsr * ((g.lrow + g.rrow) >> 2)) % 4) % 4;
    g.lds_bytes = g.nsr * (g.lrow + g.rrow) + tbl_bytes;
    const void *fn = nullptr;
    switch (n) {
    case 3: fn = mfma_ptr<3>(nb); break;
    case 5: fn = mfma_ptr<5>(nb); break;
    case 7: fn = mfma_ptr<7>(nb); break;
    case 9: fn = mfma_ptr<9>(nb); break;
    case 11: fn = mfma_ptr<11>(nb); break;
    }
    *out = g;
    return fn;
}
