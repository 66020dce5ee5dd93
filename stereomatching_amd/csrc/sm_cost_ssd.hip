// sm_cost_ssd.hip -- SSD cost mode of the hot path on the byte dot-product unit (rounds 2-4).  Since round 4 the
// plan takes the matrix-core kernel (sm_cost_mfma.hip) where both apply; this one is sm_plan_options.cost_kernel = 2
// and the second implementation the tests compare that one with.
//
// PARITY UNPINNED: the reference has no SSD implementation; the mode is
// the build's own definition (stated at the top of sm_cost.hip; the checker restates it on the CPU).
//
//   SSD_d(x, y) = sum over the n x n window of (L - R)^2 = LL(x, y) + RR(x + d, y) - 2 LR_d(x, y)
// with LL / RR the window sums of the squared left / right pixels -- one value per PIXEL and row,
// not per shift -- and LR_d the window sum of the products, the only per-shift work.  Pixels are staged as
// SIGNED bytes, pixel - 128 (differences do not care), and the row that slides out is read COMPLEMENTED on
// the right side: ~r = -r - 1, so its v_dot4 takes the products off the same accumulator (plus the old row's
// left pixels once each: a drift that is the same for every shift of a pixel, cannot change the arg-min and
// comes off `best` with LL):
//     LR = dot4(left group, ~right group of the row that slides out, LR)       NG = ceil(n/4) x v_dot4c_i32_i8
//     LR = dot4(left group, right group of the row that slides in, LR)         NG
//     -key = (LR << 9) + entry                         entry = -(RR[x + d] << 8) - offset of the entry in the lane's span
//     best = max3(best, -key of an even quad, -key of the odd quad behind it)
// The last group of a window row holds n mod 4 pixels: the other bytes of the LEFT operand are
// zeroed and their products vanish -- no correction term (as the SAD kernel needs) exists here.
//
// A lane owns PX pixels (4 apart) x 32 shifts; the right operand of shift d is the right row's dword
// at byte x - half + d + 4 g: the row is re-based to the lane's window start once, then its four byte
// alignments are cut with v_alignbyte once per row and shared by the lane's pixels and the eight shifts
// of that alignment.  RR is a table in LDS over the right-image positions of the tile, split by residue
// (see below), slid down row by row by the wave.
//
// Ghost border: as in the SAD kernel, rows / columns outside the image are staged as zero pixels in both
// images; the columns x < half (taps left of the image, where the right image is not zero) are
// recomputed by sm_cost_strip.hip behind this launch.
//
// Limits: windows up to 11 x 11 (RR - 2 LR = SSD - LL lies in (-2^23, 2^23): 24 signed bits of the key)
// and 256 shifts (the other 8).

#include "sm_internal.h"
#include "sm_cost.h"
#include <type_traits>

// signed bytes (pixel - 128, staged that way): v_dot4c_i32_i8
__device__ __forceinline__ u32 dot4(u32 a, u32 b, u32 acc) { return (u32)__builtin_amdgcn_sdot4((int)a, (int)b, (int)acc, false); }

template <int N, int PX, bool FULLD>
__global__ __launch_bounds__(64, 2) void k_ssd_dot(const u8 *__restrict__ left, const u8 *__restrict__ right,
                                                   i32 *__restrict__ web, i32 *__restrict__ best,
                                                   const SadGeom g)
{
    constexpr int HALF = N / 2, FG = N / 4, RB = N % 4, NG = FG + 1;
    constexpr u32 MASKR = RB == 1 ? 0x000000ffu : 0x00ffffffu;      // left bytes of the last group
    constexpr int NQ = 8;                                            // quads of 4 shifts per lane: 32 shifts
    constexpr int WL = NG + PX - 1;                                  // left operands of a lane's PX windows
    constexpr int K = PX + NQ + NG - 2;                              // right operands of one alignment class
    static_assert(RB == 1 || RB == 3, "odd windows");
    static_assert(N * N * 65025 < (1 << 23), "RR - 2 LR must fit 24 signed bits of the key");

    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x;
    const int pair = blockIdx.z;
    const int xw = blockIdx.x * g.tw, ty0 = blockIdx.y * g.tile_h;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *L = left + img, *R = right + img;
    const int lw = g.lrow >> 2, rw = g.rrow >> 2;                    // dwords per staged row
    u32 *sL = lds;                                                   // [nsr][lw]
    u32 *sR = sL + g.nsr * lw;                                       // [nsr][rw]
    // -(RR << 8) of the current output row by right byte position q (the window centre): negated and shifted
    // once per position where it is updated, not 44 times per lane and row where it is read.  Position q lives at
    // dword (q & 3) * rr_stride + (q >> 2): the lanes of a half-wave that differ in their shift-lane read positions
    // 32 apart, in a flat table the same bank (4-way conflicts on the 44 reads of a lane and row); split by
    // residue the four shift-lanes of a half-wave are 8 dwords apart and the four residues rr_stride = 5 mod 8
    // apart: 2-way, the floor for 32 lanes whose dword addresses span 16 banks' worth of distinct values.  (A
    // table interleaved [position][shift-lane] removes the conflicts altogether and was measured SLOWER in
    // round 3, 1.32 vs 1.22 ms: keeping it up to date takes 3 600 LDS cycles per wave and row against the 700
    // the conflicts cost, profiles/r03/ab_ssd_rr_phase.txt.  The split table is a permutation: no extra update.)
    u32 *sRR = sR + g.nsr * rw;                                      // [4][rr_stride]
    const int S = g.rr_stride;

    smc_stage_rows(lds, L, R, g, xw, ty0, HALF, tid, 0x80808080u);
    for (int r = tid; r < 4 * S; r += 64) sRR[r] = 0;
    __syncthreads();

    // ---- lane role: residue a, shift-lane sl, pixel group j
    const int a = tid & 3;
    // lane = a | j0 << 2 | sl << 3 | (rest of j): one bit of the pixel group sits BELOW the shift-lane
    // (where a wave holds two groups or more), so that a half-wave -- the unit LDS conflicts are counted in --
    // holds half as many shift-lanes: their RR reads are 32 entries apart, the same bank
    const int jb = g.nl < 16 ? 1 : 0;                   // 16 / nl pixel groups per wave
    const int sl = (tid >> (2 + jb)) & (g.nl - 1);
    const int j = ((tid >> 2) & jb) | ((tid >> (2 + jb + g.log2nl)) << jb);
    const int x0 = xw + 4 * PX * j + a;                 // pixel p of this lane: x0 + 4 p
    const int rho = (a - HALF) & 3;                     // (x - HALF) mod 4
    const int bL = (x0 - HALF - rho - (xw - g.padl)) >> 2;      // dword of the window's aligned start
    const int bR = bL + NQ * sl;                        // ... of the lane's first shift (32 sl)
    // RR entry of (pixel p, shift 32 sl + 4 m + i): position r0 + 4 (p + m) + i, r0 = x0 - (xw - padl) + 32 sl, whose
    // residue is (a + i) & 3 for every p and m: dword rrb[i] + p + m
    const int r0w = (x0 - a - (xw - g.padl) + 4 * NQ * sl) >> 2;
    int rrb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) rrb[i] = ((a + i) & 3) * S + r0w + ((a + i) >> 2);
    const int dlim = g.D - 4 * NQ * sl;                 // this lane's shifts below D

    u32 A[PX][NQ][4];            // LR window sums of (pixel, quad, shift within the quad)
    u32 LLs[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        LLs[p] = 0;             // (signed build: LL - 2 C, see the step)
#pragma unroll
        for (int m = 0; m < NQ; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) A[p][m][i] = 0;
    }

    // one window row in (rn_i), one out (ro_i; none while WARM), optionally the arg-min of row y
    auto step = [&](auto warm_tag, auto out_tag, int rn_i, int ro_i, int y) {
        constexpr bool WARM = decltype(warm_tag)::value, OUT = decltype(out_tag)::value;
        const u32 *rowLn = sL + rn_i * lw, *rowRn = sR + rn_i * rw;
        const u32 *rowLo = sL + ro_i * lw, *rowRo = sR + ro_i * rw;

        // RR: per right byte position q (the window centre), + the new row's horizontal sum of squares
        // - the old row's.  Positions whose window leaves the staged row are never read.
#ifndef SSD_EXPERIMENT_NO_RR     // (timing experiment only: how much of a row is the RR update)
        // (one position per lane and turn: 4 x 76 positions are 4.75 turns of 64 lanes; four positions per lane
        // -- the four alignments of the same dwords -- were 2 turns of which the second had 12 lanes at work)
        for (int it = tid; it < 4 * (rw - NG - 1); it += 64) {
            const int k = it >> 2, i = it & 3;          // position q = 4 k + HALF + i: its window starts at byte 4 k + i
            u32 sn = 0, so = 0;
#pragma unroll
            for (int gp = 0; gp < NG; gp++) {
                u32 v = __builtin_amdgcn_alignbyte(rowRn[k + gp + 1], rowRn[k + gp], i);
                if (gp == FG) v &= MASKR;
                sn = dot4(v, v, sn);
                if (!WARM) {
                    u32 u = __builtin_amdgcn_alignbyte(rowRo[k + gp + 1], rowRo[k + gp], i);
                    if (gp == FG) u &= MASKR;
                    so = dot4(u, u, so);
                }
            }
            const int q = 4 * k + HALF + i;
            sRR[(q & 3) * S + (q >> 2)] -= (sn - so) << 8;       // the table holds -(RR << 8): what the keys take
        }
        __syncthreads();
#endif

        // left operands of this lane's PX windows: NG groups each, 4 pixels apart -> NG + PX - 1 dwords
        u32 un[WL], unp[PX], uo[WL], uop[PX];
        {
            u32 t[WL + 1];
#pragma unroll
            for (int m = 0; m <= WL; m++) t[m] = rowLn[bL + m];
#pragma unroll
            for (int m = 0; m < WL; m++) un[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
#pragma unroll
            for (int p = 0; p < PX; p++) unp[p] = un[p + FG] & MASKR;
            if (!WARM) {
#pragma unroll
                for (int m = 0; m <= WL; m++) t[m] = rowLo[bL + m];
#pragma unroll
                for (int m = 0; m < WL; m++) uo[m] = __builtin_amdgcn_alignbyte(t[m + 1], t[m], rho);
#pragma unroll
                for (int p = 0; p < PX; p++) uop[p] = uo[p + FG] & MASKR;
            }
        }
        // LL of the lane's pixels (only `best` needs it: the arg-min does not depend on it)
        if (best) {             // uniform
#pragma unroll
        for (int p = 0; p < PX; p++) {
            u32 sn = 0, so = 0;
#pragma unroll
            for (int gp = 0; gp < NG; gp++) {
                const u32 v = gp == FG ? unp[p] : un[p + gp];
                sn = dot4(v, v, sn);
                if (!WARM) {
                    const u32 u = gp == FG ? uop[p] : uo[p + gp];
                    so = dot4(u, u, so);
                    so = dot4(u, 0x02020202u, so);      // + 2 x the old row's sum: the drift C of the LR sums, see below
                }
            }
            LLs[p] += sn - so;
        }
        }

        i32 run[PX];
#pragma unroll
        for (int p = 0; p < PX; p++) run[p] = (i32)0x80000100;      // -(0x7fffff00): "nothing yet", loses to every key
        int dl = dlim;                      // (opaque: keeps the per-lane validity tests inside the row loop)
        asm volatile("" : "+v"(dl));

        // The right row, re-based to the lane's window start once (byte offset rho, per lane): after
        // that shift 4 m + i of pixel p, group g reads dword p + m + g of the copy shifted by i more
        // bytes -- the same i for every lane, and no dword of the row is read from LDS twice.
        u32 wn[K + 1], wo[K + 1];
        {
            u32 t[K + 2];
#pragma unroll
            for (int k = 0; k < K + 2; k++) t[k] = rowRn[bR + k];
#pragma unroll
            for (int k = 0; k <= K; k++) wn[k] = __builtin_amdgcn_alignbyte(t[k + 1], t[k], rho);
            if (!WARM) {
#pragma unroll
                for (int k = 0; k < K + 2; k++) t[k] = rowRo[bR + k];
                // the old row's right operand COMPLEMENTED: as signed bytes ~r = -r - 1, so its v_dot4 SUBTRACTS the old
                // row's products (and the old row's left pixels once each, a drift C that is the same for every
                // shift of a pixel -- it cannot change the arg-min and comes off `best` through LLs)
#pragma unroll
                for (int k = 0; k < K + 2; k++) t[k] = ~t[k];
#pragma unroll
                for (int k = 0; k <= K; k++) wo[k] = __builtin_amdgcn_alignbyte(t[k + 1], t[k], rho);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u32 rn[K], ro[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                rn[k] = i ? __builtin_amdgcn_alignbyte(wn[k + 1], wn[k], i) : wn[k];
                if (!WARM) ro[k] = i ? __builtin_amdgcn_alignbyte(wo[k + 1], wo[k], i) : wo[k];
            }
            // RR of (pixel p, shift 4 m + i) is entry r0 + 4 (p + m) + i: a window of PX entries slides over m
            // -(RR << 8) of the window's entries, each MINUS its own offset 4 k + i from the lane's first entry
            // (k = p + m): that offset is the shift 4 m + i of pixel p plus 4 p, so the low byte of a key tells the
            // shift apart as before, and the subtraction is done once per entry (PX + NQ - 1 of them) instead of
            // once per (pixel, shift); the 4 p come off with the shift-lane's base at the end of the row
            u32 nrr[PX + 1];
            if (OUT) {
#pragma unroll
                for (int p = 0; p < PX; p++) nrr[p] = sRR[rrb[i] + p] - (u32)(4 * p + i);
            }
            i32 held[PX];               // the keys of an even quad wait for the odd one's: one v_max3_i32 for two
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NQ; m++) {
                if (OUT && m + 1 < NQ) nrr[PX] = sRR[rrb[i] + PX + m] - (u32)(4 * (PX + m) + i);
                // the PX pixels' chains side by side, group by group: a v_dot4 that accumulates onto the one
                // issued just before it costs a wait state (three before any other reader), and the compiler
                // pads with s_nop what the source order does not separate
                u32 acc[PX];
#pragma unroll
                for (int p = 0; p < PX; p++) acc[p] = A[p][m][i];
                // one chain per pixel: the old row's groups against the complemented right row take their products off
                if (!WARM) {
#pragma unroll
                    for (int gp = 0; gp < NG; gp++)
#pragma unroll
                        for (int p = 0; p < PX; p++) acc[p] = dot4(gp == FG ? uop[p] : uo[p + gp], ro[p + m + gp], acc[p]);
                }
#pragma unroll
                for (int gp = 0; gp < NG; gp++)
#pragma unroll
                    for (int p = 0; p < PX; p++) acc[p] = dot4(gp == FG ? unp[p] : un[p + gp], rn[p + m + gp], acc[p]);
#pragma unroll
                for (int p = 0; p < PX; p++) A[p][m][i] = acc[p];
                if (OUT) {
                    // key = (RR - 2 LR) << 8 | shift within the lane, signed: the smallest wins, i.e. the lowest
                    // SSD (LL is the same for all shifts of a pixel) and among equals the first shift.  Formed
                    // NEGATED, -key = (LR << 9) + (-(RR << 8) - shift), one v_lshl_add_u32 behind one subtract of a
                    // constant from the (negated, pre-shifted) RR entry, and the MAXIMUM is kept.  (Plain C on
                    // purpose: as an inline-asm v_mad_i32_i24 this read a v_dot4 result without the wait states
                    // the compiler gives its own instructions -- wrong first rows of every tile.)
#pragma unroll
                    for (int p = 0; p < PX; p++) {
                        i32 nkey = (i32)((acc[p] << 9) + nrr[p]);
                        // FULLD: the lanes' 32 shifts each are all below D (D = 32 x shift-lanes); otherwise the
                        // last shift-lane holds shifts >= D, which must never win
                        if (!FULLD && 4 * m + i >= dl) nkey = (i32)0x80000100;
                        if (m & 1) run[p] = max(max(run[p], held[p]), nkey);
                        else held[p] = nkey;
                    }
                    if (m & 1) {
#pragma unroll
                        for (int p = 0; p < PX; p++)
                            asm volatile("" : : "v"(run[p]));   // (a use here: the maxima are otherwise deferred to the
                                                                // row's end and every key kept alive until then)
                    }
                }
                if (OUT) {
#pragma unroll
                    for (int p = 0; p < PX; p++) nrr[p] = nrr[p + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (OUT) {
#pragma unroll
            for (int p = 0; p < PX; p++) {
                i32 key = 4 * NQ * sl - 4 * p - run[p];     // back to the key; its low 8 bits become the shift itself
                for (int k = 0; k < g.log2nl; k++) key = min(key, __shfl_xor(key, (4 << jb) << k));
                const int x = x0 + 4 * p;
                if (sl == 0 && x < g.w) {
                    const size_t o = ((size_t)pair * g.h + y) * g.w + x;
                    web[o] = (key & 255) + 1;
                    if (best) best[o] = (key >> 8) + (i32)LLs[p];
                }
            }
        }
        __syncthreads();            // RR is updated by the next step
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
static const void *ssd_ptr(int px, bool fulld)
{
    if (px == 2) return fulld ? (const void *)k_ssd_dot<N, 2, true> : (const void *)k_ssd_dot<N, 2, false>;
    if (px == 4) return fulld ? (const void *)k_ssd_dot<N, 4, true> : (const void *)k_ssd_dot<N, 4, false>;
    return nullptr;
}

// fills g and returns the kernel, or nullptr if this shape is not built (caller falls back)
const void *sm_ssd_dot_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *out)
{
    SadGeom g;
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts; g.waves = 1;
    const int half = plan->square_width / 2, n = 2 * half + 1;
    g.ghost = plan->border == SM_GHOST;
    if (n < 3 || n > 11 || g.D > 256 || plan->opt.cost_kernel == 1) return nullptr;
    const int nql = 8;
    int px = 4;             // (2: narrower tiles, more set-up per pixel; measured slower at every BASELINE configuration)
    if (plan->opt.cost_pixels_per_lane == 2 || plan->opt.cost_pixels_per_lane == 4) px = plan->opt.cost_pixels_per_lane;
    g.nl = 1; g.log2nl = 0;
    while (g.nl * 4 * nql < g.D) { g.nl <<= 1; g.log2nl++; }
    g.tw = 4 * px * (16 / g.nl);
    g.tiles_x = (g.w + g.tw - 1) / g.tw;
    const int ng = n / 4 + 1;
    g.padl = 4 * ((half + 3 + 3) / 4);
    // left row: dwords bL .. bL + NG + PX - 1; right: bR + 1 + (PX + NQ + NG - 2) of the last shift-lane
    g.lrow = 8 * ((g.padl + g.tw + 4 * (ng + 1) + 7) / 8);
    g.rrow = 8 * ((g.padl + g.tw + 4 * (g.nl * nql + ng + 3) + 7) / 8);
    g.q_tail = g.D - 4 * nql * (g.nl - 1);          // shifts of the last shift-lane below D
    g.q_last = nql - 1;
    g.tbl_pad = 0;
    g.rr_stride = g.rrow / 4 + 1;
    while (g.rr_stride % 8 != 5) g.rr_stride++;
    const size_t rr_bytes = 16 * (size_t)g.rr_stride;
    const int slots = 256 * 4 * 2;
    int best_th = 0; double best_cost = 0;
    for (int th = 8; th <= 128; th += 4) {
        const size_t lds = (size_t)(th + n - 1) * (g.lrow + g.rrow) + rr_bytes;
        if (lds > 160 * 1024 / 8) break;
        const long long tiles = (long long)g.tiles_x * ((g.h + th - 1) / th) * pairs;
        const long long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * (th + 0.45 * (n - 1) + 2.0);
        if (!best_th || cost < best_cost) { best_th = th; best_cost = cost; }
    }
    if (!best_th) return nullptr;
    if (plan->opt.cost_tile_h > 0) {         // an explicit tile height, clamped to what a workgroup's LDS holds
        best_th = plan->opt.cost_tile_h;
        while (best_th > 1 && (size_t)(best_th + n - 1) * (g.lrow + g.rrow) + rr_bytes > 64 * 1024) best_th--;
    }
    g.tile_h = best_th < g.h ? best_th : g.h;
    g.tiles_y = (g.h + g.tile_h - 1) / g.tile_h;
    g.nsr = g.tile_h + n - 1;
    g.fast_stage = g.w % 4 == 0 && ((uintptr_t)d_left & 3) == 0 && ((uintptr_t)d_right & 3) == 0 &&
                   g.lrow + g.rrow <= 4 * 256;
    g.lds_bytes = g.nsr * (g.lrow + g.rrow) + (int)rr_bytes;
    g.nql = nql; g.px = px;
    const bool fulld = g.D == 4 * nql * g.nl;
    const void *fn = nullptr;
    switch (n) {
    case 3: fn = ssd_ptr<3>(px, fulld); break;
    case 5: fn = ssd_ptr<5>(px, fulld); break;
    case 7: fn = ssd_ptr<7>(px, fulld); break;
    case 9: fn = ssd_ptr<9>(px, fulld); break;
    case 11: fn = ssd_ptr<11>(px, fulld); break;
    }
    *out = g;
    return fn;
}
