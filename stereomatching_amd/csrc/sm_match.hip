// sm_match.hip -- the hot path (match cost -> S x S window sum -> masked
// score -> winner-take-all over the shifts, one fused launch): plan geometry,
// the general POPCOUNT kernels and the launch switch.  The common windows run
// on the bit-sliced kernel of sm_match_bs.hip (about 2x faster); the kernels in
// this file cover every other window up to 25 x 25 and D up to 1024, and the
// generic kernel at the bottom everything beyond.
//
// Replaces, for all D shifts at once (paths relative to /root/reference):
//   fillup_matches                src/stereo.cu:127-137   (src/stereo.c:113-127)
//   addup_pixels_in_square        src/stereo.cu:142-155   (src/stereo.c:132-148)
//   record_score / fillup_scores  src/stereo.cu:185-207   (src/stereo.c:172-192)
//   find_highest_scoring_shifts   src/stereo.cu:211-225   (src/stereo.c:196-220)
// Closed form:
//   m_d(x,y) = [L(x,y) == R(x+d, y)]
//   A_d(x,y) = sum over the n x n window of m_d          (n = 2*(S/2)+1)
//   s_d      = m_d ? A_d : 0
//   best     = max_d s_d ;  web = 1 + max{ d : s_d == best }
// None of the D match planes (u8) or score planes (i32) ever exists in HBM.
//
// Design (gfx950, wave64; integer VALU + LDS, no MFMA):
//  * Input is the packed ext image (sm_internal.h): 1 bit per pixel with the
//    border rule already applied, so rows are plain coalesced dword loads.
//  * A workgroup owns a tile of tw x tile_h pixels.  It stages the tile's
//    n-1+tile_h rows (window halo included) of L and R bits in LDS twice:
//    plain, and "spread" (bit i of a row moved to bit 2i).
//  * A lane owns a run of P = 8 consecutive pixels and DSET = 16 consecutive
//    shifts and marches down the tile keeping the 128 window sums A[d][j] in
//    registers.  The shift range of a run is split over nl = D/16 adjacent
//    lanes; their winners are merged with DPP row operations.
//  * Sliding the window down by one row needs  + popcount(new row's window)
//    - popcount(old row's window).  The old row is stored complemented, so
//    the update is  + popcount(new) + popcount(~old) - n, and because spread
//    rows interleave (new -> even bits, old -> odd bits) both popcounts are
//    ONE v_bfe_u32 + ONE v_bcnt_u32_b32 on a word
//           Z_d = IL ^ (IR >> 2d)
//    (interleaving commutes with XOR and turns a shift by d into a shift by
//    2d, so it is done once per row in LDS, never per shift).  The "- n" is
//    dropped: every shift of a pixel carries the same bias, which cannot
//    change the arg-max, and is subtracted again when `best` is written.
//  * Winner key = (A << 10 | d) masked by the centre match bit; the unsigned
//    max of the keys is "highest score, then highest shift", the reference's
//    last-wins rule.  A key of 0 means no shift matched -> web = D, best = 0
//    (all scores 0, the last shift wins; src/stereo.c:211-218).
//
// Three instantiations by window size: A (n <= 9: Z fits 32 bits), B (n <= 16)
// and C (n <= 25) use 64-bit Z.  Larger windows or D > 1024 take the generic
// kernel at the bottom (correct for every input, not tuned).

#include "sm_internal.h"

#include <algorithm>
#include <cmath>
#include <stdlib.h>
#include <string.h>

static_assert(SM_DSET == 16 && SM_P == 8, "pixel_winner / the key loop are written out for 16 x 8");

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ u32 spread16(u32 x)
{
    // bit i -> bit 2i, for the low 16 bits
    x &= 0xffffu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);   // ((hi:lo) >> (sh & 31)) low 32
}
template <int CTRL>
__device__ __forceinline__ i32 dpp_max(i32 v)
{
    i32 o = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
    return v > o ? v : o;
}

// all-reduce max of P keys over groups of nl adjacent lanes (nl uniform,
// power of two): one uniform branch per level, not per key
__device__ __forceinline__ void group_max(i32 (&v)[SM_P], int nl)
{
    if (nl < 2) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) v[j] = dpp_max<0xB1>(v[j]);    // quad_perm [1,0,3,2]
    if (nl < 4) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) v[j] = dpp_max<0x4E>(v[j]);    // quad_perm [2,3,0,1]
    if (nl < 8) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) v[j] = dpp_max<0x141>(v[j]);   // row_half_mirror
    if (nl < 16) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) v[j] = dpp_max<0x140>(v[j]);   // row_mirror
    if (nl < 32) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) { i32 o = __shfl_xor(v[j], 16); v[j] = v[j] > o ? v[j] : o; }
    if (nl < 64) return;
#pragma unroll
    for (int j = 0; j < SM_P; j++) { i32 o = __shfl_xor(v[j], 32); v[j] = v[j] > o ? v[j] : o; }
}

// Measured issue rates on gfx950 (tools/ubench_valu*.hip): v_and/or/xor/add/
// sub/lshr/ashr/bitop3 retire a wave64 in ~2 cycles, but v_bfe/bcnt/alignbit/
// max/max3/lshl/lshl_or/mad/cmp take ~4.  The winner search is written for that:
// the centre-match bit of the candidate is kept in the SIGN bit of r (one slow
// left shift per pixel, then r += r per candidate, a full-rate op), an
// unmatched candidate gets the sign bit forced on (one bitop3) and the keys
// are compared as signed ints, so unmatched < every matched key.
//   per candidate: v_lshl_or (slow) + v_bitop3 + v_add (fast) + half a v_max3_i32
__device__ __forceinline__ i32 imax(i32 a, i32 b) { return a > b ? a : b; }

// (x << 10) | dd as ONE v_lshl_or_b32, and x + x as a v_add_u32: left to
// itself hipcc emits v_lshlrev + an extra v_bitop3 for the first and a
// (half-rate) v_lshlrev for the second.
template <int DD>
__device__ __forceinline__ u32 make_key(u32 a)
{
    u32 k;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(k) : "v"(a), "n"(SM_KEY_DBITS), "n"(DD));
    return k;
}
__device__ __forceinline__ u32 twice(u32 r)
{
    u32 o;
    asm("v_add_u32 %0, %1, %1" : "=v"(o) : "v"(r));
    return o;
}

template <int J, bool FULLD>
__device__ __forceinline__ i32 pixel_winner(const u32 (&A)[SM_DSET][SM_P], u32 lc, u32 rc, u32 vb)
{
    // bit (J+dd) of rj = centre match of pixel J at shift d0+dd
    u32 rj = ~(rc ^ (u32)__builtin_amdgcn_sbfe((int)lc, J, 1));
    if (!FULLD) rj &= vb << J;
    u32 r = rj << (31 - (J + SM_DSET - 1));     // candidate dd = 15 in the sign bit
    i32 k = (i32)0x80000000u;
#define SM_CAND(DD, R) (i32)(make_key<DD>(A[DD][J]) | (~(R) & 0x80000000u))
#define SM_PAIR(HI)                                                             \
    {                                                                           \
        const u32 r1 = twice(r);                                                \
        k = imax(imax(SM_CAND(HI, r), SM_CAND(HI - 1, r1)), k);                 \
        r = twice(r1);                                                          \
    }
    SM_PAIR(15) SM_PAIR(13) SM_PAIR(11) SM_PAIR(9) SM_PAIR(7) SM_PAIR(5) SM_PAIR(3) SM_PAIR(1)
#undef SM_PAIR
#undef SM_CAND
    return k;
}

// ---------------------------------------------------------------------------
// the tiled kernel
// ---------------------------------------------------------------------------

template <int MODE, bool FULLD, bool GHOST>
__global__ __launch_bounds__(256) void k_match_wta(const u32 *__restrict__ ext,
                                                   i32 *__restrict__ web,
                                                   i32 *__restrict__ best,
                                                   const MatchGeom g)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    constexpr int ZW = MODE == SM_KERNEL_A ? 1 : 2;   // words of Z
    constexpr int RW = ZW + 1;                          // words of the IR base

    const int tid = threadIdx.x;
    const int pair = blockIdx.z;
    int tile_x, tile_y;
    sm_xcd_tile(g.tiles_x, g.tiles_y, tile_x, tile_y);
    const int tx0 = tile_x * g.tw;
    const int ty0 = tile_y * g.tile_h;
    const int n = g.n, half = g.half;
    const int plw = g.plw, prw = g.prw, nsr = g.nsr;

    const u32 *extL = ext + (size_t)pair * 2 * g.ext_image_words;
    const u32 *extR = extL + g.ext_image_words;

    u32 *pL = lds;                 // plain rows, left   [nsr][plw]
    u32 *pR = pL + nsr * plw;      // plain rows, right  [nsr][prw]
    u32 *sL = pR + nsr * prw;      // spread rows, left  [nsr][2*plw]
    u32 *sR = sL + nsr * 2 * plw;  // spread rows, right [nsr][2*prw]

    // ---- stage the tile (+ window halo) into LDS: coalesced dword row loads
    {
        const int wx0 = tx0 >> 5;   // pad_l == SM_PADT, so tile words are aligned
        const int per_row = plw + prw;
        for (int it = tid; it < nsr * per_row; it += blockDim.x) {
            const int row = it / per_row;
            const int k = it - row * per_row;
            const bool is_r = k >= plw;
            const int kk = is_r ? k - plw : k;
            const u32 v = (is_r ? extR : extL)[(size_t)(ty0 + row) * g.ext_words + wx0 + kk];
            const u32 lo = spread16(v), hi = spread16(v >> 16);
            if (is_r) {
                pR[row * prw + kk] = v;
                sR[row * 2 * prw + 2 * kk] = lo;
                sR[row * 2 * prw + 2 * kk + 1] = hi;
            } else {
                pL[row * plw + kk] = v;
                sL[row * 2 * plw + 2 * kk] = lo;
                sL[row * 2 * plw + 2 * kk + 1] = hi;
            }
        }
    }
    __syncthreads();

    // ---- lane role
    const int s = tid & (g.nl - 1);       // which 16 shifts
    const int r = tid >> g.log2nl;        // which run of 8 pixels
    const int x0l = r * SM_P;
    const int d0 = s * SM_DSET;
    const int x0 = tx0 + x0l;

    // bit positions inside a staged row (bit = SM_PADT + tile-local x)
    const int bL = SM_PADT + x0l - half;  // window start, left
    const int bR = bL + d0;               // window start, right, shift d0
    const int wLs = (2 * bL) >> 5, shLs = (2 * bL) & 31;   // in spread rows
    const int wRs = (2 * bR) >> 5, shRs = (2 * bR) & 31;
    const int bLc = SM_PADT + x0l, bRc = bLc + d0;         // centre bits, plain rows
    const int wLc = bLc >> 5, shLc = bLc & 31;
    const int wRc = bRc >> 5, shRc = bRc & 31;

    // masks for the window extraction
    const u32 w2n = 2u * (u32)n;
    u32 mask_a[SM_P];                     // MODE A: 2n bits from bit 2j (wave-uniform)
    if (MODE == SM_KERNEL_A) {
#pragma unroll
        for (int j = 0; j < SM_P; j++) mask_a[j] = ((w2n >= 32 ? 0u : (1u << w2n)) - 1u) << (2 * j);
    }
    u32 mask_b = 0;                       // MODE B: low 2n bits
    u32 mask_c[SM_P];                     // MODE C: low 2j+2n-32 bits of Z1
    if (MODE == SM_KERNEL_B)
        mask_b = w2n >= 32 ? 0xffffffffu : ((1u << w2n) - 1u);
    if (MODE == SM_KERNEL_C) {
#pragma unroll
        for (int j = 0; j < SM_P; j++) {
            const u32 wj = 2 * j + w2n - 32;
            mask_c[j] = wj >= 32 ? 0xffffffffu : ((1u << wj) - 1u);
        }
    }

    // ghost border: which window columns lie inside the image (per lane,
    // constant over rows), spread like the data
    u32 cv_e[ZW];
    if (GHOST) {
        u32 cv = 0;
        for (int i = 0; i < 16 * ZW; i++) {
            const int x = x0 - half + i;
            if (x >= 0 && x < g.w) cv |= 1u << i;
        }
        cv_e[0] = spread16(cv);
        if (ZW == 2) cv_e[1] = spread16(cv >> 16);
    }

    // shifts >= D contribute nothing: clear their centre-match bits
    u32 vb = 0xffffu;
    if (!FULLD) {
        const int dlim = g.D - d0;
        vb = dlim >= SM_DSET ? 0xffffu : (dlim <= 0 ? 0u : ((1u << dlim) - 1u));
    }

    u32 A[SM_DSET][SM_P];
#pragma unroll
    for (int dd = 0; dd < SM_DSET; dd++)
#pragma unroll
        for (int j = 0; j < SM_P; j++)
            A[dd][j] = 0;

    const int rows_out = min(g.tile_h, g.h - ty0);   // >= 1
    const int steps = rows_out + n - 1;

    for (int e = 0; e < steps; e++) {
        // ---- window rows for this step: new = staged row e, old = e - n
        u32 il[ZW], ir[RW];
        {
            const u32 *rowL = sL + e * 2 * plw + wLs;
            const u32 *rowR = sR + e * 2 * prw + wRs;
#pragma unroll
            for (int k = 0; k < ZW; k++) il[k] = alignbit(rowL[k + 1], rowL[k], shLs);
#pragma unroll
            for (int k = 0; k < RW; k++) ir[k] = alignbit(rowR[k + 1], rowR[k], shRs);
        }
        if (e >= n) {
            const u32 *rowL = sL + (e - n) * 2 * plw + wLs;
            const u32 *rowR = sR + (e - n) * 2 * prw + wRs;
#pragma unroll
            for (int k = 0; k < ZW; k++) il[k] |= alignbit(rowL[k + 1], rowL[k], shLs) << 1;
#pragma unroll
            for (int k = 0; k < RW; k++) ir[k] |= alignbit(rowR[k + 1], rowR[k], shRs) << 1;
        }
        // even (new) bits: flip so that 1 = match; odd (old) bits: 1 = mismatch
#pragma unroll
        for (int k = 0; k < ZW; k++) il[k] ^= 0x55555555u;

        u32 vz[ZW];
        if (GHOST) {
            const int y_new = ty0 - half + e, y_old = y_new - n;
            const bool v_new = y_new >= 0 && y_new < g.h;
            const bool v_old = e >= n && y_old >= 0 && y_old < g.h;
#pragma unroll
            for (int k = 0; k < ZW; k++)
                vz[k] = (v_new ? cv_e[k] : 0u) | (v_old ? (cv_e[k] << 1) : 0u);
        }

        // ---- accumulate: one bfe + one bcnt per (pixel, shift)
#pragma unroll
        for (int dd = 0; dd < SM_DSET; dd++) {
            u32 z[ZW];
#pragma unroll
            for (int k = 0; k < ZW; k++) {
                z[k] = il[k] ^ alignbit(ir[k + 1], ir[k], 2 * dd);
                if (GHOST) z[k] &= vz[k];
            }
#pragma unroll
            for (int j = 0; j < SM_P; j++) {
                if (MODE == SM_KERNEL_A) {
                    A[dd][j] += __builtin_popcount(z[0] & mask_a[j]);   // v_and (fast) + v_bcnt
                } else if (MODE == SM_KERNEL_B) {
                    const u32 t = j ? alignbit(z[ZW - 1], z[0], 2 * j) : z[0];
                    A[dd][j] += __builtin_popcount(t & mask_b);
                } else {
                    A[dd][j] += __builtin_popcount(z[0] >> (2 * j));
                    A[dd][j] += __builtin_popcount(z[ZW - 1] & mask_c[j]);
                }
            }
        }

        // ---- winner-take-all for output row t
        if (e >= n - 1) {
            const int t = e - (n - 1);
            const int y = ty0 + t;
            const u32 *rowLc = pL + (t + half) * plw + wLc;
            const u32 *rowRc = pR + (t + half) * prw + wRc;
            const u32 lc = alignbit(rowLc[1], rowLc[0], shLc);
            const u32 rc = alignbit(rowRc[1], rowRc[0], shRc);

            // signed keys: negative = no shift of this lane matched
            i32 key[SM_P];
            key[0] = pixel_winner<0, FULLD>(A, lc, rc, vb) + d0;
            key[1] = pixel_winner<1, FULLD>(A, lc, rc, vb) + d0;
            key[2] = pixel_winner<2, FULLD>(A, lc, rc, vb) + d0;
            key[3] = pixel_winner<3, FULLD>(A, lc, rc, vb) + d0;
            key[4] = pixel_winner<4, FULLD>(A, lc, rc, vb) + d0;
            key[5] = pixel_winner<5, FULLD>(A, lc, rc, vb) + d0;
            key[6] = pixel_winner<6, FULLD>(A, lc, rc, vb) + d0;
            key[7] = pixel_winner<7, FULLD>(A, lc, rc, vb) + d0;
            group_max(key, g.nl);

            if (s == 0 && x0 < g.w) {
                // bias carried by every A of pixel j at row t
                int rows_in = t;          // old rows removed so far that were valid
                if (GHOST) {
                    const int a = max(0, ty0 - half), b = min(g.h, ty0 - half + t);
                    rows_in = max(0, b - a);
                }
                i32 wv[SM_P], bv[SM_P];
#pragma unroll
                for (int j = 0; j < SM_P; j++) {
                    // a matched shift always scores >= 1 (its own pixel is in the window)
                    const u32 sc = key[j] < 0 ? 0u : (u32)key[j] >> SM_KEY_DBITS;
                    int cols_in = n;
                    if (GHOST) {
                        const int x = x0 + j;
                        cols_in = min(g.w - 1, x + half) - max(0, x - half) + 1;
                    }
                    wv[j] = sc ? (i32)((u32)key[j] & ((1u << SM_KEY_DBITS) - 1u)) + 1 : g.D;
                    bv[j] = sc ? (i32)sc - rows_in * cols_in : 0;
                }
                const size_t o = ((size_t)pair * g.h + y) * g.w + x0;
                if (g.vec_ok && x0 + SM_P <= g.w) {
                    int4 *pw = reinterpret_cast<int4 *>(web + o);
                    pw[0] = make_int4(wv[0], wv[1], wv[2], wv[3]);
                    pw[1] = make_int4(wv[4], wv[5], wv[6], wv[7]);
                    if (best) {
                        int4 *pb = reinterpret_cast<int4 *>(best + o);
                        pb[0] = make_int4(bv[0], bv[1], bv[2], bv[3]);
                        pb[1] = make_int4(bv[4], bv[5], bv[6], bv[7]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < SM_P; j++) {
                        if (x0 + j < g.w) {
                            web[o + j] = wv[j];
                            if (best) best[o + j] = bv[j];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// generic kernel: any window, any D.  One lane per pixel, direct window sums
// from the ext image in global memory.  O(n * n/32) per matched shift.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_match_wta_generic(const u32 *__restrict__ ext,
                                                           i32 *__restrict__ web,
                                                           i32 *__restrict__ best,
                                                           const MatchGeom g, int ghost)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int pair = blockIdx.z;
    if (x >= g.w) return;
    const u32 *extL = ext + (size_t)pair * 2 * g.ext_image_words;
    const u32 *extR = extL + g.ext_image_words;
    const int half = g.half;

    auto bit = [&](const u32 *img, int xx, int yy) -> u32 {
        const int b = xx + g.pad_l;
        return (img[(size_t)(yy + half) * g.ext_words + (b >> 5)] >> (b & 31)) & 1u;
    };
    auto word = [&](const u32 *img, int xx, int yy) -> u32 {   // 32 bits from xx
        const int b = xx + g.pad_l;
        const u32 *p = img + (size_t)(yy + half) * g.ext_words + (b >> 5);
        return alignbit(p[1], p[0], b & 31);
    };

    // window extent; in ghost mode taps outside the image count 0
    int xa = x - half, xb = x + half, ya = y - half, yb = y + half;
    if (ghost) {
        xa = max(xa, 0); xb = min(xb, g.w - 1);
        ya = max(ya, 0); yb = min(yb, g.h - 1);
    }

    const u32 lc = bit(extL, x, y);
    i32 bs = 0, bw = g.D;
    for (int d = 0; d < g.D; d++) {
        if (bit(extR, x + d, y) != lc) continue;   // no match at the pixel: score 0
        i32 sum = 0;
        for (int yy = ya; yy <= yb; yy++) {
            for (int xs = xa; xs <= xb; xs += 32) {
                const int cnt = min(32, xb - xs + 1);
                u32 m = ~(word(extL, xs, yy) ^ word(extR, xs + d, yy));
                if (cnt < 32) m &= (1u << cnt) - 1u;
                sum += __builtin_popcount(m);
            }
        }
        if (sum >= bs) { bs = sum; bw = d + 1; }
    }
    const size_t o = ((size_t)pair * g.h + y) * g.w + x;
    web[o] = bw;
    if (best) best[o] = bs;
}

// ---------------------------------------------------------------------------
// host side: geometry and launch
// ---------------------------------------------------------------------------

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

static bool tiled_fulld(const MatchGeom &g) { return g.nl * g.ds == g.D; }

static const void *tiled_kernel_ptr(int mode, bool fulld, bool ghost)
{
#define SM_ROW(M)                                                                       \
    fulld ? (ghost ? (const void *)k_match_wta<M, true, true> : (const void *)k_match_wta<M, true, false>) \
          : (ghost ? (const void *)k_match_wta<M, false, true> : (const void *)k_match_wta<M, false, false>)
    switch (mode) {
    case SM_KERNEL_A: return SM_ROW(SM_KERNEL_A);
    case SM_KERNEL_B: return SM_ROW(SM_KERNEL_B);
    default: return SM_ROW(SM_KERNEL_C);
    }
#undef SM_ROW
}

int sm_match_configure(sm_plan *plan)
{
    MatchGeom &g = plan->g;
    const int W = plan->width, H = plan->height, D = plan->num_shifts;
    g.w = W; g.h = H; g.D = D;
    g.half = plan->square_width / 2;
    g.n = 2 * g.half + 1;

    int kernel;
    if (D > (1 << SM_KEY_DBITS) || g.n > 25) kernel = SM_KERNEL_GENERIC;
    else if (g.n <= 9) kernel = SM_KERNEL_A;
    else if (g.n <= 16) kernel = SM_KERNEL_B;
    else kernel = SM_KERNEL_C;
    // the bit-sliced kernel where it is built (common windows, D <= 512);
    // sm_plan_options.kernel_family = 1 keeps the general kernels (A/B testing)
    const bool ghost = plan->border == SM_GHOST;
    auto nl_for = [&](int ds, int &log2nl) { int nl = 1; log2nl = 0; while (nl * ds < D) { nl <<= 1; log2nl++; } return nl; };
    {
        const bool want_bs = plan->opt.kernel_family != 1;
        int l2;
        const int ds0 = sm_bs_default_ds(g.n);
        if (want_bs && kernel != SM_KERNEL_GENERIC && ds0 && nl_for(ds0, l2) <= 32)
            kernel = SM_KERNEL_BS;
    }
    plan->kernel = kernel;

    if (kernel == SM_KERNEL_GENERIC) {
        g.pad_l = 32 * ceil_div(std::max(g.half, 1), 32);
        // word() reads 2 words starting at bit x+d+pad_l with x <= W-1+half
        g.ext_words = (g.pad_l + W + g.half + D + 31) / 32 + 2;
        g.ext_rows = H + 2 * g.half;
        g.ext_image_words = (long long)g.ext_words * g.ext_rows;
        g.tile_h = g.tw = g.runs = g.nl = g.log2nl = g.threads = g.ds = 0;
        g.plw = g.prw = g.nsr = g.tiles_x = g.tiles_y = g.vec_ok = g.lds_bytes = g.cap2 = g.duo = 0;
        g.prio_pattern = 0;
        g.prio_unit = 14;
        g.xmerge = g.xm_off = g.xm_words = 0;
        g.prio_shift = 0;
        g.prio_on_change = 0;
        g.edge_words_l = g.edge_words_r = g.ext_words;
        snprintf(plan->describe, sizeof plan->describe,
                 "generic kernel (n=%d, D=%d): 1 lane/pixel, direct window sums", g.n, D);
        return SM_OK;
    }

    g.pad_l = SM_PADT;
    const bool bs = kernel == SM_KERNEL_BS;
    int cus = 256, th_env = 0, ds_env = 0;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, plan->device) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        th_env = plan->opt.tile_h;              // sm_plan_create_ex: tuning / tests only
        ds_env = plan->opt.shifts_per_lane;
    }

    // ---- geometry for `ds` shifts per lane, with the tile height chosen by a cost
    // model.  Tall tiles amortise the n-1 warm-up rows, but the grid should fill the
    // chip's resident slots in whole rounds: a tail round with a third of the CUs busy
    // costs as much as a full one.  Model: a workgroup puts threads/256 waves on each
    // SIMD; a SIMD issues one wave-instruction per 2 cycles, a single wave at most one
    // per 4; rounds run back to back; the work of a lane-row is proportional to ds.
    auto configure = [&](int ds, MatchGeom &o, int &rows_words_o, bool duo = false) -> double {
        o = g;
        o.ds = ds;
        o.duo = duo ? 1 : 0;
        o.nl = nl_for(ds, o.log2nl);
        int rows_words;
        if (bs) {
            // one wave per workgroup: 64/nl words of 32 pixels, nl shift-lanes each
            o.runs = 64 / o.nl;
            o.threads = duo ? 128 : 64;     // duo: two waves, the upper and the lower half of the tile
            o.tw = o.runs * 32;
            o.plw = o.runs + 2;
            o.prw = o.runs + (o.nl * ds + 31) / 32 + 4;
            rows_words = o.plw + o.prw;
        } else {
            o.runs = o.nl == 1 ? 64 : (o.nl <= 8 ? 32 : 256 / o.nl);
            o.threads = o.runs * o.nl;
            o.tw = o.runs * SM_P;
            o.plw = (SM_PADT + o.tw + o.half + 31) / 32 + 1;
            o.prw = (SM_PADT + o.tw + o.half + o.nl * ds + 31) / 32 + 1;
            rows_words = (o.plw + o.prw) * 3;     // plain + spread
        }
        rows_words_o = rows_words;
        o.tiles_x = ceil_div(W, o.tw);
        const bool fulld = o.nl * ds == D;
        const void *kfn = bs ? sm_bs_kernel_ptr(o.n, ds, fulld, ghost, false, duo) : tiled_kernel_ptr(kernel, fulld, ghost);
        const double wps = o.threads / 256.0;            // waves per SIMD per workgroup
        // warm-up rows are cheaper than output rows (no arg-max, no output); the constant is
        // the per-workgroup overhead (staging, lane set-up) in output-row units.  Refit on
        // same-device tile-height sweeps (tools/tune_tile_h.py, C2 / C3 / C4 x 8).
        // (duo: half the window rows + 1, and the exchange of the partial sums)
        const double warm = duo ? 0.42 * (o.half + 1) + 2.3
                          : bs ? 0.42 * (o.n - 1) + 1.8 : 0.4 * (o.n - 1) + 1.0;
        // lane-row work relative to ds = 16: the per-row shared views and one more merge level

        // duo: 2 * th rows per workgroup, and behind the staged rows the exchange block
        // [2 halves of the shifts][ds / 2 * SB / 2 plane pairs][64 lanes] of 8 bytes
        int sb = 0;
        while ((1 << sb) <= o.n * o.n) sb++;
        const int rows_per_wg = duo ? 2 : 1;
        // lane merge through LDS (k_match_bs, g.xmerge): where at least 4 lanes share a word; per wave
        // 4 x NPG blocks of 1 KB behind the staged rows, in a two-wave workgroup over the exchange slots
        const int ab = ds == 16 ? 4 : ds == 8 ? 3 : 2;
        const int npg = (sb + ab + 3) / 4;
        // Taken where it pays: 16 shifts per lane and at least 8 lanes per word (C3: -4.3 % of the launch's
        // VALU instructions, -3 % of its time; C5: -10 %).  With 4 lanes per word there are only two DPP levels
        // to save and the batch's bursts of stores cost more than that (C4 x 8: +5 %); the 8-shifts-per-lane
        // builds run 4- to 9-row tiles, whose last batch is mostly empty.  lane_merge = 2 forces it wherever
        // it is possible (tests, measurements), 1 forbids it.  profiles/r04/ab_lane_merge.txt
        const bool xm_possible = bs && o.log2nl >= 2 && o.nl <= 32;
        o.xmerge = xm_possible && plan->opt.lane_merge != 1 &&
                   (plan->opt.lane_merge == 2 || ((ds == 16 || ds == 4) && o.log2nl >= 3));
        // lane-row work relative to ds = 16: the per-row shared views and the merge levels weigh more the fewer
        // shifts a lane carries (fitted to same-device timings: profiles/r02/ds8_small_grids_sweep.txt, r04/ab_ds4.txt)
        // (round 4, tools/ds_choice_check.py over 14 shapes: 8 shifts per lane was the best of the three ONCE and
        // was chosen seven times -- its weight went from 0.55 to 0.65, the LDS-merged 4-shift build's from 0.36 to
        // 0.33, and a 16-shift row whose lanes are merged through LDS counts 0.95)
        const double work = ds == 16 ? (o.xmerge ? 0.95 : 1.0) : ds == 8 ? 0.5 * 1.30
                          : o.xmerge ? 0.25 * 1.32 : 0.25 * (1.30 + 0.15 * o.log2nl);
        const int mb_words = o.xmerge ? npg * 1024 : 0;
        auto lds_words = [&](int th, int &xm_off) {
            const int staged = ((rows_per_wg * th + o.n - 1) * rows_words + 3) & ~3;
            if (!bs) { xm_off = 0; return (rows_per_wg * th + o.n - 1) * rows_words; }
            if (duo) {
                const int slot = ds * sb * 32;                    // words of one exchange slot
                const int half = std::max(slot, mb_words);
                xm_off = staged + half;
                return staged + 2 * half;
            }
            xm_off = staged;
            return staged + mb_words;
        };
        auto lds_of = [&](int th) { int off; return lds_words(th, off) * 4; };
        int th = 0;
        double best_cost = 0;
        for (int c = 2; c <= 256; c++) {
            if (c > H && c != 2) break;
            const int cand = std::min(c, H);
            if (lds_of(cand) > 64 * 1024) break;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, o.threads, lds_of(cand)) != hipSuccess
                || per_cu < 1)
                per_cu = 1;
            const long long tiles = (long long)o.tiles_x * ceil_div(H, rows_per_wg * cand) * plan->max_pairs;
            const long long slots = (long long)cus * per_cu;
            double cost = 0;
            for (long long left = tiles; left > 0; left -= slots) {
                const long long m = std::min(left, slots);
                // the busiest SIMD of this round hosts j waves; measured: one wave alone
                // retires an instruction every ~5.5 cycles (popcount kernels; ~4.3 for the
                // bit-sliced kernel), two co-resident waves ~5.5 each, beyond that they
                // share ~2.5 cycles/instr
                const long long wg_per_cu = (m + cus - 1) / cus;
                const int j = std::max(1, (int)std::ceil((double)wg_per_cu * wps - 1e-9));
                const double cpi = bs ? (j <= 1 ? 4.3 : std::max(5.5, 2.5 * j)) : std::max(5.5, 2.5 * j);
                cost += (cand + warm) * work * cpi;
            }
            if (th == 0 || cost < best_cost * 0.999) { th = cand; best_cost = cost; }
        }
        if (th == 0) th = 1;
        if (th_env > 0) th = std::min(th_env, H);
        while (lds_of(th) > 64 * 1024 && th > 1) th--;
        o.tile_h = th;
        o.tiles_y = ceil_div(H, rows_per_wg * th);
        o.nsr = rows_per_wg * th + o.n - 1;
        o.lds_bytes = lds_words(th, o.xm_off) * 4;
        o.xm_words = mb_words;
        // A grid that fits the chip in one round must also be SPREAD evenly: where the
        // registers allow more resident workgroups than the round needs (7x7: 3 waves
        // per SIMD, 2 needed) the dispatcher may stack 3 waves on some SIMDs and leave
        // others with 1, and the launch then lasts as long as the crowded ones (measured
        // at 8 x 1080p: 89 us spread evenly, 117 us not).  Two caps:
        //  * per SIMD: a grid that fits at two waves per SIMD launches the kernel's
        //    two-wave variant (k_match_bs<..., CAP2>), where one exists;
        //  * per CU: an LDS request larger than the tile needs -- LDS per workgroup in
        //    (160 KB / (cap + 1), 160 KB / cap] admits exactly `cap` workgroups per CU.
        o.cap2 = 0;
        if (bs) {
            const long long tiles = (long long)o.tiles_x * o.tiles_y * plan->max_pairs;
            const int cap = (int)((tiles + cus - 1) / cus);
            const void *kcap = sm_bs_kernel_ptr(o.n, ds, fulld, ghost, true, duo);
            const void *kuse = kfn;
            if (kcap && cap <= 8 && !plan->opt.no_two_wave_cap) { o.cap2 = 1; kuse = kcap; }
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kuse, o.threads, o.lds_bytes) == hipSuccess
                && cap >= 2 && cap < per_cu) {
                const int lds_cu = 160 * 1024, granule = 1280;
                int want = std::min(64 * 1024, lds_cu / cap / granule * granule);
                int got = 0;
                if (want > o.lds_bytes &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, kuse, o.threads, want) == hipSuccess &&
                    got == cap)
                    o.lds_bytes = want;
            }
        }
        o.ext_words = (o.tiles_x - 1) * (o.tw / 32) + o.prw;
        o.ext_rows = o.tiles_y * rows_per_wg * th + o.n - 1;
        o.ext_image_words = (long long)o.ext_words * o.ext_rows;
        o.vec_ok = (W % 4) == 0;
        return best_cost;
    };

    // shifts per lane: 16 for the popcount kernels; for the bit-sliced kernel what is
    // built for this window (16 where it exists: measured faster than 8, fewer shared
    // views and merge levels), sm_plan_options.shifts_per_lane overrides for tuning
    int ds = 16;
    MatchGeom gsel;
    int rws = 0;
    // Two-wave workgroups with a shared warm-up (k_match_bs<..., DUO>): HALF + 1 warm-up rows
    // per wave instead of N, for an exchange through LDS.  Measured on one device at the same
    // tile height (tools/ab_duo.sh): C3 95.3 -> 92.1 us, C4 x 8 83.6 -> 80.5, C5 197.9 -> 185.9,
    // C2 20.5 -> 19.4, 21 x 21 at 4K 88.5 -> 77.4, C1 9.9 -> 10.0: taken wherever the cost
    // model says so (sm_plan_options.workgroup_waves overrides: tuning, tests).
    int duo_env = -1;
    if (plan->opt.workgroup_waves) duo_env = plan->opt.workgroup_waves == 2;
    auto duo_built = [&](int d) {
        int l2;
        return sm_bs_kernel_ptr(g.n, d, nl_for(d, l2) * d == D, ghost, false, true) != nullptr;
    };
    // lower-cost geometry of the two workgroup shapes for `d` shifts per lane
    auto configure_best = [&](int d, MatchGeom &o, int &r) -> double {
        const bool can = bs && duo_built(d);
        double c1 = 0, c2 = 0;
        MatchGeom o2;
        int r2 = 0;
        if (!(can && duo_env == 1)) c1 = configure(d, o, r, false);
        if (can && duo_env != 0) {
            c2 = configure(d, o2, r2, true);
            if (duo_env == 1 || c2 < c1) { o = o2; r = r2; return c2; }
        }
        return c1;
    };
    if (bs) {
        ds = sm_bs_default_ds(g.n);
        int l2;
        const bool has8 = sm_bs_kernel_ptr(g.n, 8, true, ghost, false) && nl_for(8, l2) <= 32;
        if ((ds_env == 4 || ds_env == 8 || ds_env == 16) && sm_bs_kernel_ptr(g.n, ds_env, true, ghost, false) &&
            nl_for(ds_env, l2) <= 32) {
            ds = ds_env;
        } else {
            // A grid that leaves most SIMDs with ONE wave (a single 1080p pair at 16 shifts per lane: 864
            // workgroups) runs at the rate of a lone wave; with 8 -- or 4 -- shifts per lane the same job is
            // more workgroups of less work each, on narrower tiles that can be taller for the same number of
            // waves (less warm-up per output row).  The cost model decides, with 5 % in favour of the wider
            // lane.  Measured: C2 29.6 (16) -> 19.1 (8) -> 16.5 us (4, lanes merged through LDS), C1 18.4 ->
            // 9.8 -> 7.2 us; the full-chip configurations stay at 16 (profiles/r04/ab_ds4.txt).
            // What the model cannot see -- the narrow lanes win by latency hiding on grids that leave the chip
            // partly empty, not by instruction count -- is put in as a rule taken from tools/ds_choice_check.py
            // (14 shapes, profiles/r04/ds_choice_*.txt): below 0.3 G pixel-shifts per launch, or for windows of
            // 13 x 13 and more (their warm-up weighs less on narrow, tall tiles), all three are candidates; above
            // it a window that has a 16-shift build takes it (the worst miss of this rule: 5 %).
            const double pxshifts = (double)W * H * D * plan->max_pairs;
            const bool small_or_tall = pxshifts <= 0.3e9 || g.n >= 13;
            const bool has4 = plan->opt.no_four_shift_lanes == 0 && small_or_tall &&
                              sm_bs_kernel_ptr(g.n, 4, true, ghost, false) && nl_for(4, l2) <= 32;
            double cbest = 0;
            int dbest = 0;
            for (int d : {16, 8, 4}) {
                if (d == 16 && ds != 16) continue;          // (windows whose 16-shift build does not exist)
                if (d == 8 && (!has8 || (ds == 16 && !small_or_tall))) continue;
                if (d == 4 && !has4) continue;
                MatchGeom gd;
                int rd = 0;
                const double c = configure_best(d, gd, rd);
                if (!dbest || c < 0.95 * cbest) { dbest = d; cbest = c; }
            }
            if (dbest) ds = dbest;
        }
    }
    configure_best(ds, gsel, rws);
    g = gsel;
#ifndef SM_EDGE_TRIM
#define SM_EDGE_TRIM 1      // 0: edges for every ext column (same-device A/B builds)
#endif
    g.edge_words_l = g.edge_words_r = g.ext_words;
    if (SM_EDGE_TRIM) {
        g.edge_words_l = std::min(g.ext_words, (g.pad_l + W + g.half - 1) / 32 + 1);
        g.edge_words_r = std::min(g.ext_words, (g.pad_l + W + g.half + D - 2) / 32 + 1);
    }
    g.prio_pattern = sm_bs_default_pattern(g.duo != 0);
    g.prio_unit = 14;
    // The launches of the 8- and 4-shifts-per-lane builds are short (a single small pair: 16 us; the reference's
    // defaults at 4K: 58 us): with slices of four units (~31 us) the favoured wave of a SIMD pair never or hardly
    // changes, and the other one finishes alone.  They swap every unit (~8 us): C2's match launch 17.1 -> 15.6 us,
    // step -4 %; 21 x 21 / 30 shifts at 4K -1..2 %; the long launches (16 shifts per lane) lose 2-3 % with it and
    // keep the four-unit slices (profiles/r04/ab_prio_unit.txt).
    if (bs && g.ds < 16) g.prio_pattern = 0xAAAAAAAAu;
    if (plan->opt.priority_pattern) g.prio_pattern = plan->opt.priority_pattern;   // tuning
    if (plan->opt.priority_unit_log2 >= 8 && plan->opt.priority_unit_log2 <= 20) g.prio_unit = plan->opt.priority_unit_log2;
    // the bit that tells a SIMD's two waves apart (see the kernel): the wave slot.  The workgroup's slot on its
    // CU (priority_class 2) wins 1-2 % (6 % at 21 x 21) when the match launch follows ITSELF, as in a timing loop
    // of match launches -- and LOSES 5 % in the real step, where it follows the edge kernel and its workgroups
    // find other slots (tools/sustained_ab.sh, profiles/r03/sustained_ab.txt): an option for tuning, not the default
    g.prio_shift = 0;
    if (plan->opt.priority_class == 2) {
        // the workgroup's slot tells a SIMD's two waves apart only for two-wave workgroups of a launch that fits
        // the chip in ONE round (later workgroups land in whatever slot is free): anything else keeps the wave slot
        const long long tiles = (long long)g.tiles_x * g.tiles_y * plan->max_pairs;
        if (bs && g.duo && tiles <= 4ll * cus) g.prio_shift = 16;                           // tuning
    }
    g.prio_on_change = plan->opt.priority_on_change == 1;                                 // tuning (default: once per row)

    snprintf(plan->describe, sizeof plan->describe,
             "%s (n=%d, D=%d, %s): tile %dx%d px, %d threads "
             "(%d runs x %d shift-lanes of %d), grid %dx%d, LDS %d B/wg%s%s%s, ext %dx%d words",
             bs ? "bit-sliced kernel" : kernel == SM_KERNEL_A ? "tiled kernel A"
                : kernel == SM_KERNEL_B ? "tiled kernel B" : "tiled kernel C",
             g.n, D, ghost ? "ghost" : "toroidal",
             g.tw, g.duo ? 2 * g.tile_h : g.tile_h, g.threads, g.runs, g.nl, g.ds, g.tiles_x, g.tiles_y, g.lds_bytes,
             g.duo ? ", two-wave workgroups" : g.cap2 ? ", 2 waves/SIMD variant" : "",
             g.xmerge ? ", lanes merged through LDS" : "",
             g.prio_shift ? ", favoured by workgroup slot"
                          : plan->opt.priority_class == 2 ? ", priority class 2 not applicable: by wave slot" : "",
             g.ext_words, g.ext_rows);
    return SM_OK;
}

template <int MODE>
static void launch_tiled(const sm_plan *plan, const MatchGeom &g, int pairs, i32 *d_web, i32 *d_best, hipStream_t st)
{
    const dim3 grid(g.tiles_x, g.tiles_y, pairs), block(g.threads);
    const bool fulld = tiled_fulld(g);
    const bool ghost = plan->border == SM_GHOST;
#define SM_GO(F, G) \
    hipLaunchKernelGGL((k_match_wta<MODE, F, G>), grid, block, g.lds_bytes, st, plan->d_ext, d_web, d_best, g)
    if (fulld) { if (ghost) SM_GO(true, true); else SM_GO(true, false); }
    else       { if (ghost) SM_GO(false, true); else SM_GO(false, false); }
#undef SM_GO
}

int sm_match_launch(const sm_plan *plan, const MatchLaunch &l, int pairs, i32 *d_web, i32 *d_best, hipStream_t st)
{
    const MatchGeom &g = l.g;
    switch (plan->kernel) {
    case SM_KERNEL_BS: return sm_bs_launch(plan, l, pairs, d_web, d_best, st);
    case SM_KERNEL_A: launch_tiled<SM_KERNEL_A>(plan, g, pairs, d_web, d_best, st); break;
    case SM_KERNEL_B: launch_tiled<SM_KERNEL_B>(plan, g, pairs, d_web, d_best, st); break;
    case SM_KERNEL_C: launch_tiled<SM_KERNEL_C>(plan, g, pairs, d_web, d_best, st); break;
    default: {
        const dim3 grid(ceil_div(g.w, 256), g.h, pairs), block(256);
        hipLaunchKernelGGL(k_match_wta_generic, grid, block, 0, st, plan->d_ext, d_web, d_best,
                           g, plan->border == SM_GHOST ? 1 : 0);
    }
    }
    SM_LAUNCH_CHECK("k_match_wta");
    return SM_OK;
}
