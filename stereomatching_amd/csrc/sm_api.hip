// sm_api.hip -- C-ABI entry points (include/stereo_hip.h) and the small
// kernels around the hot path: edge detection straight into the packed ext
// image, u8 -> ext packing, the debug tap, and step 3 (hole fill, min/max,
// contour).  The hot path itself is sm_match.hip.

#include "sm_internal.h"

#include <limits.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------

static thread_local char g_err[512] = "";

int sm_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *sm_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// step 1: edges, written directly in the hot path's packed format
// ---------------------------------------------------------------------------

// The 3-vs-3 contrast test of src/stereo.c:19-27 on integer side sums in
// units of 1/256 (brightness is k/256.0, src/image.c:9-15; the ghost halo
// 128.0 is 32768).  A three-term sum of such values is exact in double, so
// (a+b+c)/3.0 == sum/768.0 with a single rounding, and every later operation
// is one IEEE operation exactly as in the C source.  Built with
// -ffp-contract=off; tests/test_hip_gpu.py (test_edge_decision_exhaustive) checks
// all 766*766 in-image sum pairs against the host's arithmetic.
__device__ __forceinline__ bool contrast_test(int sa, int sb, double threshold)
{
    const double ma = (double)sa / 768.0;
    const double mb = (double)sb / 768.0;
    const double overall = (ma + mb) / 2.0;
    double limit = threshold * overall;
    limit = limit > 0.0 ? limit : 0.0;
    limit = limit < 1.0 ? limit : 1.0;
    return fabs(ma - mb) > limit;
}

__device__ __forceinline__ int pos_mod(int v, int m)
{
    int r = v % m;
    return r < 0 ? r + m : r;
}

// edge value of image pixel (x, y), 0 <= x < w, 0 <= y < h
__device__ __forceinline__ u32 edge_at(const u8 *__restrict__ gray, int w, int h, int x, int y,
                                       double threshold, bool ghost)
{
    int v[3][3];
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            int xx = x + dx, yy = y + dy;
            int val;
            if (ghost) {
                const bool in = xx >= 0 && xx < w && yy >= 0 && yy < h;
                val = in ? gray[(size_t)yy * w + xx] : 32768;
            } else {
                xx = xx < 0 ? w - 1 : (xx >= w ? 0 : xx);
                yy = yy < 0 ? h - 1 : (yy >= h ? 0 : yy);
                val = gray[(size_t)yy * w + xx];
            }
            v[dy + 1][dx + 1] = val;
        }
    }
    // v[row][col]: row 0 = y-1, col 0 = x-1
    // left | right                       src/stereo.c:16-28
    if (contrast_test(v[0][0] + v[1][0] + v[2][0], v[0][2] + v[1][2] + v[2][2], threshold)) return 1;
    // top | bottom                       src/stereo.c:30-42
    if (contrast_test(v[0][0] + v[0][1] + v[0][2], v[2][0] + v[2][1] + v[2][2], threshold)) return 1;
    // up-left | down-right               src/stereo.c:44-56
    if (contrast_test(v[0][0] + v[0][1] + v[1][0], v[1][2] + v[2][1] + v[2][2], threshold)) return 1;
    // down-left | up-right               src/stereo.c:58-70
    if (contrast_test(v[2][0] + v[2][1] + v[1][0], v[0][1] + v[0][2] + v[1][2], threshold)) return 1;
    return 0;
}

__global__ void k_edge_table(double threshold, u8 *__restrict__ table)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x, sa = blockIdx.y;
    if (sb < 766) table[sa * 766 + sb] = contrast_test(sa, sb, threshold);
}

// Per-threshold decision tables.  For a fixed left sum sa the exact test is
// true for right sums sb <= lo(sa) and sb >= hi(sa) and false in between: with
// sb moving away from sa, |ma - mb| grows by 1/768 per unit and the limit
// threshold*(ma+mb)/2 by at most 1/1536, so the difference is monotone by a
// margin of ~1e-3, far above the rounding of the double operations.  The
// tables are BUILT with the exact double test (one workgroup per sa evaluates
// all 766 sb) and the threshold form is VERIFIED while building: if any row is
// not "true prefix, false middle, true suffix", bad_flag is raised and the
// edge kernel keeps using the double arithmetic.  The edge kernel then needs
// two integer compares per orientation instead of two double divisions.
__global__ __launch_bounds__(256) void k_edge_thresholds(double threshold, u32 *__restrict__ tab,
                                                         i32 *__restrict__ bad_flag)
{
    __shared__ int lo, hi, n_lo, n_hi;
    const int sa = blockIdx.x;
    if (threadIdx.x == 0) { lo = -1; hi = 766; n_lo = 0; n_hi = 0; }
    __syncthreads();
    for (int sb = threadIdx.x; sb < 766; sb += blockDim.x) {
        if (contrast_test(sa, sb, threshold)) {
            if (sb <= sa) { atomicMax(&lo, sb); atomicAdd(&n_lo, 1); }
            if (sb >= sa) { atomicMin(&hi, sb); atomicAdd(&n_hi, 1); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tab[sa] = (u32)(lo & 0xffff) | ((u32)hi << 16);   // lo = -1 -> 0xffff (never <=)
        if (n_lo != lo + 1 || n_hi != 766 - hi) atomicOr(bad_flag, 1);
    }
}

// f32 prefilter in front of the tables.  In exact arithmetic the test is
//     E = |sa - sb| - theta * (sa + sb) > 0,   theta = threshold / 2
// (the clamp to [0,1] never binds for in-image sums).  The sums are integers
// below 2^11, exact in f32; with T = (float)theta,
//     F = fma(sa + sb, -T, |sa - sb|)                        (one rounding)
// differs from E by at most 1530 * |T - theta| <= 1530 * 2^-26 < 2.3e-5 plus the
// fma rounding, which is <= 2^-25 whenever |F| < 1.  So |F| > 2^-12 (2.4e-4)
// leaves a real margin > 2e-4 sum units -- a relative margin > 1e-7 on
// quantities the double evaluation gets right to ~1e-15: the sign of F IS the
// double decision.  Only |F| <= 2^-12 (the few sum pairs next to the boundary)
// consults the table.  Same function in the edge kernels and in the exhaustive
// debug table, so the test covers what runs.  Three full-rate VALU operations
// per orientation (the abs and the negation are source modifiers).
#define SM_EDGE_MARGIN 0.000244140625f
__device__ __forceinline__ float edge_delta(float sa, float sb, float neg_t)
{
    return __builtin_fmaf(sa + sb, neg_t, __builtin_fabsf(sa - sb));
}
__device__ __forceinline__ bool edge_from_table(const u32 *tab, int sa, int sb)
{
    const u32 lh = tab[sa];
    return sb <= (int)(short)(lh & 0xffff) || sb >= (int)(lh >> 16);
}

__global__ void k_edge_table_fast(const u32 *__restrict__ tab, float neg_t, u8 *__restrict__ table)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x, sa = blockIdx.y;
    if (sb >= 766) return;
    const float delta = edge_delta((float)sa, (float)sb, neg_t);
    table[sa * 766 + sb] = delta > SM_EDGE_MARGIN ? 1 : delta < -SM_EDGE_MARGIN ? 0
                                                      : edge_from_table(tab, sa, sb);
}

#define SM_EDGE_ROWS 32   // ext rows one wave walks down

// byte B of a dword as f32 (v_cvt_f32_ubyteB)
template <int B> __device__ __forceinline__ float cvt_ubyte(u32 q)
{
    float f;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(q));
    if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(q));
    if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(q));
    if (B == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(q));
    return f;
}

// One pixel's decision from its 8 orientation sums (f32, exact integers).
// sa/sb order: left|right, top|bottom, up-left|down-right, down-left|up-right
// (src/stereo.c:16-70).  `exact` (ghost pixels on or outside the image border,
// whose sums contain the 128.0 halo and are outside the tables; or no usable
// tables at all) takes the double arithmetic of the reference.
// `known_edge`: a ghost-mode pixel ON the image border of an image at least 2 x 2.  One of
// its axis-aligned tests has three halo pixels (128.0 each) on one side and only in-image
// pixels (< 1.0 each) on the other -- or, at a corner, two halo pixels more on one side than
// on the other -- so the side means differ by more than 40 while the limit is clamped to
// [0, 1] (src/stereo-ghost.c:18-30): it is an edge for every threshold, and no arithmetic
// is spent on it (the double path it used to take made ghost-mode edges 4x slower).
template <bool TABLES>
__device__ __forceinline__ u32 edge_decide(const float (&sa)[4], const float (&sb)[4],
                                           const u32 *__restrict__ tab, double threshold,
                                           float neg_t, bool exact, bool known_edge = false)
{
    u32 e;
    if (TABLES) {
        float dl[4];
#pragma unroll
        for (int o = 0; o < 4; o++) dl[o] = edge_delta(sa[o], sb[o], neg_t);
        const float dmax = fmaxf(fmaxf(dl[0], dl[1]), fmaxf(dl[2], dl[3]));
        e = __float_as_uint(SM_EDGE_MARGIN - dmax) >> 31;                 // dmax > margin
        // rare: the deciding sum pair is next to the boundary -> ask the table
        // (never with halo sums: they are not table indices)
        if (!exact && !known_edge && __builtin_fabsf(dmax) <= SM_EDGE_MARGIN) {
#pragma unroll
            for (int o = 0; o < 4; o++)
                if (dl[o] >= -SM_EDGE_MARGIN)
                    e |= edge_from_table(tab, (int)sa[o], (int)sb[o]) ? 1u : 0u;
        }
    }
    if (!TABLES || exact) {
        e = 0;
#pragma unroll
        for (int o = 0; o < 4; o++)
            e |= contrast_test((int)sa[o], (int)sb[o], threshold) ? 1u : 0u;
    }
    return known_edge ? 1u : e;
}

// Edge detection straight into the packed ext image.  No LDS, no barrier: a
// wave owns a strip of 64 ext pixels x SM_EDGE_ROWS ext rows and walks down it;
// each lane keeps the 3 x 3 gray neighbourhood of its pixel in registers and
// loads three bytes (x-1, x, x+1) of the next row per step.  The wave's 64
// decisions become two ext words via ballot.  Border rule at load time: wrapped
// coordinates (toroidal) or the 128.0 halo, 32768 in units of 1/256 (ghost).
// This is the any-width kernel; widths that are a multiple of 4 take
// k_edges_ext4 below.
template <bool GHOST, bool TABLES>
__global__ __launch_bounds__(256) void k_edges_ext(const u8 *__restrict__ src_l,
                                                   const u8 *__restrict__ src_r,
                                                   u8 *__restrict__ edges_l,
                                                   u8 *__restrict__ edges_r,
                                                   u32 *__restrict__ ext,
                                                   const u32 *__restrict__ tab,
                                                   const MatchGeom g, double threshold, float neg_t)
{
    const int tid = threadIdx.x;
    const int xe = blockIdx.x * 256 + tid;
    const int ye0 = blockIdx.y * SM_EDGE_ROWS;
    const int pair = blockIdx.z >> 1, side = blockIdx.z & 1;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *src = (side ? src_r : src_l) + img;
    u8 *edges = side ? edges_r : edges_l;

    const int x = xe - g.pad_l;
    const u32 in_x = (xe < g.ext_words * 32 && (!GHOST || (x >= 0 && x < g.w))) ? 1u : 0u;
    const bool store_x = x >= 0 && x < g.w && edges != nullptr;
    const bool inner_x = x > 0 && x < g.w - 1;             // ghost: no halo in the 3 columns
    u32 *ext_img = ext + (size_t)blockIdx.z * g.ext_rows * g.ext_words;
    const int wd = xe >> 5;

    // source columns of x-1, x, x+1
    int xc[3];
    bool vx[3];
    if (GHOST) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int xx = x - 1 + k;
            vx[k] = xx >= 0 && xx < g.w;
            xc[k] = vx[k] ? xx : 0;
        }
    } else {
        xc[1] = pos_mod(x, g.w);
        xc[0] = xc[1] == 0 ? g.w - 1 : xc[1] - 1;
        xc[2] = xc[1] + 1 == g.w ? 0 : xc[1] + 1;
        vx[0] = vx[1] = vx[2] = true;
    }

    // rows: image row of ext row ye is ye - half; the walk starts one above
    int y_img = ye0 - g.half - 1;
    int ys = GHOST ? y_img : pos_mod(y_img, g.h);       // source row (toroidal: wrapped)
    auto load_row = [&](float (&o)[3]) {
        const bool vy = !GHOST || (y_img >= 0 && y_img < g.h);
        const u8 *row = src + (size_t)(vy ? ys : 0) * g.w;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = (float)row[xc[k]];
            o[k] = (vy && vx[k]) ? v : 32768.0f;
        }
        y_img++;
        ys = GHOST ? y_img : (ys + 1 == g.h ? 0 : ys + 1);
    };

    float v[3][3];          // v[row][col]: row 0 = y-1, col 0 = x-1
    load_row(v[0]);
    load_row(v[1]);
    const int rows = min(SM_EDGE_ROWS, g.ext_rows - ye0);
    for (int rr = 0; rr < rows; rr++) {
        load_row(v[2]);
        const int ye = ye0 + rr;
        const int y = ye - g.half;
        const bool in_y = y >= 0 && y < g.h;         // uniform
        const float sa[4] = {v[0][0] + v[1][0] + v[2][0],      // left      src/stereo.c:16-28
                             v[0][0] + v[0][1] + v[0][2],      // top       src/stereo.c:30-42
                             v[0][0] + v[0][1] + v[1][0],      // up-left   src/stereo.c:44-56
                             v[2][0] + v[2][1] + v[1][0]};     // down-left src/stereo.c:58-70
        const float sb[4] = {v[0][2] + v[1][2] + v[2][2],      // right
                             v[2][0] + v[2][1] + v[2][2],      // bottom
                             v[1][2] + v[2][1] + v[2][2],      // down-right
                             v[0][1] + v[0][2] + v[1][2]};     // up-right
        const bool on_border = GHOST && !(inner_x && y > 0 && y < g.h - 1);
        const bool big = g.w >= 2 && g.h >= 2;         // uniform
        const bool exact = on_border && !(TABLES && big);
        const u32 e = edge_decide<TABLES>(sa, sb, tab, threshold, neg_t, exact, on_border && TABLES && big);
        const u32 val = e & in_x & ((!GHOST || in_y) ? 1u : 0u);
        if (store_x && in_y) edges[img + (size_t)y * g.w + x] = (u8)val;
        const unsigned long long bal = __ballot(val != 0);
        if ((tid & 63) == 0) {
            u32 *row = ext_img + (size_t)ye * g.ext_words;
            if (wd < g.ext_words) row[wd] = (u32)bal;
            if (wd + 1 < g.ext_words) row[wd + 1] = (u32)(bal >> 32);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { v[0][k] = v[1][k]; v[1][k] = v[2][k]; }
    }
}

// Same decision, FOUR pixels per lane (images whose width is a multiple of 4).
// Per row a lane loads ONE aligned dword (its 4 gray values); the pixels left
// and right of the quad are bytes of the neighbouring lanes' dwords, fetched
// with DPP wave shifts -- only lane 0 / lane 63 of a wave need a real byte load
// (one instruction serves both).  A wave's whole strip is SM_EDGE4_ROWS + 2
// rows: ALL its loads are issued before the first decision (the row loop is
// unrolled over a compile-time row count), so a wave pays the HBM latency once
// instead of once per row -- the row-at-a-time version of this kernel spent half
// its wave-cycles waiting.  Sums are shared: per row the 5 pair sums and 4 triple
// sums of horizontally adjacent pixels are formed once and carried down (the
// bottom sums of row y are the top sums of row y+2); column sums serve as `left`
// of one pixel and `right` of another.  A lane's 4 decisions form a nibble; 8
// adjacent lanes OR their nibbles together (DPP) into one ext word.
#ifndef SM_EDGE4_ROWS
#define SM_EDGE4_ROWS 4
#endif
template <bool GHOST, bool TABLES, bool STACKED = false>
__global__ __launch_bounds__(256) void k_edges_ext4(const u8 *__restrict__ src_l,
                                                    const u8 *__restrict__ src_r,
                                                    u8 *__restrict__ edges_l,
                                                    u8 *__restrict__ edges_r,
                                                    u32 *__restrict__ ext,
                                                    const u32 *__restrict__ tab,
                                                    const MatchGeom g, double threshold, float neg_t)
{
    constexpr int R = SM_EDGE4_ROWS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // The workgroup's four waves lie side by side in one strip or (STACKED) take four strips
    // below each other, 256 ext pixels wide: the host picks the second where the round-up
    // of a 1024-pixel workgroup along x would leave much of the launch idle (a 1080p ext row
    // is 560 lanes: 9 waves instead of 12; 8 x 1080p: 40.8 -> 34.0 us; C5: 29.5 -> 26.1) and
    // the first where a row is whole workgroups anyway (4K toroidal: 17.0 vs 18.1 us stacked).
    // first of this lane's 4 ext pixels, first ext row of the wave's strip
    const int xe = STACKED ? (blockIdx.x * 64 + lane) * 4 : (blockIdx.x * 256 + tid) * 4;
    const int ye0 = STACKED ? (blockIdx.y * 4 + (tid >> 6)) * R : blockIdx.y * R;
    if (STACKED && ye0 >= g.ext_rows) return;             // the round-up of the strips (wave-uniform)
    // columns no valid output pixel can reach (the match kernel's tile round-up; for the left image
    // also the shift range): left as they are (wave-uniform).  Zero since the plan was created, or -- after an
    // sm_load_edges, whose k_pack_ext writes every ext column -- stale content of that call: either way no stored
    // output pixel reads them (tests: sm_load_edges, then sm_find_edges and the match launch on one plan)
    if (((xe - 4 * lane) >> 5) >= ((blockIdx.z & 1) ? g.edge_words_r : g.edge_words_l)) return;
    const int pair = blockIdx.z >> 1, side = blockIdx.z & 1;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *src = (side ? src_r : src_l) + img;
    u8 *edges = side ? edges_r : edges_l;

    const int x = xe - g.pad_l;                            // multiple of 4
    const bool in_ext = xe < g.ext_words * 32;
    const bool quad_in = x >= 0 && x < g.w;                // all 4 inside (w % 4 == 0)
    const bool inner_x = x > 0 && x + 4 < g.w;             // ghost: no halo in the 6 columns
    u32 *ext_img = ext + (size_t)blockIdx.z * g.ext_rows * g.ext_words;
    const int wd = xe >> 5;

    // Ghost mode: a wave whose strip, with its one-pixel ring of neighbours, lies strictly inside the
    // image (x in [1, w - 2], y in [1, h - 2]) meets no halo, no border pixel and no round-up: it runs
    // the body without a single validity select (`SEL` false) -- at 4K that is 97 % of the waves; the
    // others keep the selects.  Toroidal mode has no selects to begin with.
    auto body = [&](auto sel_tag) {
        constexpr bool SEL = decltype(sel_tag)::value;
        // source columns: the aligned quad; the single pixel left (lane 0) or right
        // (lane 63) of the wave's span -- the other lanes' value of `xn` is unused
        int xq, xn;
        bool vq, vl, vr;
        if (GHOST && !SEL) {
            vq = vl = vr = true;
            xq = x;
            xn = lane == 63 ? x + 4 : x - 1;
        } else if (GHOST) {
            vq = quad_in; vl = x - 1 >= 0 && x - 1 < g.w; vr = x + 4 >= 0 && x + 4 < g.w;
            xq = vq ? x : 0;
            xn = lane == 63 ? (vr ? x + 4 : 0) : (vl ? x - 1 : 0);
        } else {
            // x is in [-pad_l, ext width - pad_l): one conditional add or subtract wraps it
            // whenever the image is at least as wide as either pad (the usual case);
            // the division is the fallback for images narrower than their padding
            const int over = g.ext_words * 32 - g.pad_l - g.w;     // uniform: columns right of the image
            if (g.w >= g.pad_l && g.w >= over) xq = x < 0 ? x + g.w : (x >= g.w ? x - g.w : x);
            else                               xq = pos_mod(x, g.w);
            // lanes of the grid's round-up beyond the ext image load nothing meaningful, but
            // they do load: keep their addresses inside the row (one wrap is not enough there)
            if (!in_ext) xq = 0;
            xn = lane == 63 ? (xq + 4 == g.w ? 0 : xq + 4) : (xq == 0 ? g.w - 1 : xq - 1);
            vq = vl = vr = true;
        }

        // all loads of the strip: rows ye0-half-1 ... ye0-half+R (border rule on the row)
        u32 q4[R + 2], nb[R + 2];
        bool vy[R + 2];
        {
            int y_img = ye0 - g.half - 1;
            // wrapped source row of the strip's first row: y_img >= -half - 1 >= -h always; one
            // conditional add or subtract covers up to 2h, the division (a few dozen scalar
            // instructions per wave, on the CU's one scalar unit: 6 % of this kernel's time) is
            // left for the round-up rows of very small images
            int ys = y_img;
            if (!GHOST) ys = y_img < 0 ? y_img + g.h : (y_img < g.h ? y_img : (y_img < 2 * g.h ? y_img - g.h : pos_mod(y_img, g.h)));
    #pragma unroll
            for (int k = 0; k < R + 2; k++) {
                vy[k] = !SEL || (y_img >= 0 && y_img < g.h);
                const u8 *row = src + (size_t)(vy[k] ? ys : 0) * g.w;
                q4[k] = *reinterpret_cast<const u32 *>(row + xq);
                nb[k] = row[xn];
                y_img++;
                ys = GHOST ? y_img : (ys + 1 == g.h ? 0 : ys + 1);
            }
        }
        // gray values of a row as f32 (col 0 = x-1 ... col 5 = x+4), its pair and triple sums
        auto unpack_row = [&](int k, float (&o)[6], float (&p)[5], float (&s3)[4]) {
            const u32 q = q4[k];
            // lane i-1's / lane i+1's dword (wave_shr:1 / wave_shl:1)
            const u32 from_l = (u32)__builtin_amdgcn_update_dpp(0, (int)q, 0x138, 0xf, 0xf, false);
            const u32 from_r = (u32)__builtin_amdgcn_update_dpp(0, (int)q, 0x130, 0xf, 0xf, false);
            const u32 lq = lane == 0 ? nb[k] << 24 : from_l;
            const u32 rq = lane == 63 ? nb[k] : from_r;
            // v_cvt_f32_ubyteN: byte -> f32 in one instruction, and opaque to the
            // optimiser (plain casts get their f32 sums folded back into integer adds
            // plus one conversion per SUM, which is more work)
            const float g0 = cvt_ubyte<0>(q), g1 = cvt_ubyte<1>(q),
                        g2 = cvt_ubyte<2>(q), g3 = cvt_ubyte<3>(q);
            const bool okq = vy[k] && vq;
            o[0] = (vy[k] && vl) ? cvt_ubyte<3>(lq) : 32768.0f;
            o[1] = okq ? g0 : 32768.0f;
            o[2] = okq ? g1 : 32768.0f;
            o[3] = okq ? g2 : 32768.0f;
            o[4] = okq ? g3 : 32768.0f;
            o[5] = (vy[k] && vr) ? cvt_ubyte<0>(rq) : 32768.0f;
    #pragma unroll
            for (int c = 0; c < 5; c++) p[c] = o[c] + o[c + 1];
    #pragma unroll
            for (int c = 0; c < 4; c++) s3[c] = p[c] + o[c + 2];
        };

        float v[3][6], p[3][5], s3[3][4];      // [row][col]: row 0 = y-1
        unpack_row(0, v[0], p[0], s3[0]);
        unpack_row(1, v[1], p[1], s3[1]);
    #pragma unroll
        for (int rr = 0; rr < R; rr++) {
            unpack_row(rr + 2, v[2], p[2], s3[2]);
            const int ye = ye0 + rr;
            const int y = ye - g.half;
            const bool in_y = y >= 0 && y < g.h;         // uniform
            // ghost: pixels on the image border are edges by construction (edge_decide); only
            // images narrower or lower than 2 keep the double path for them
            const bool big = g.w >= 2 && g.h >= 2;                       // uniform
            const bool row_border = SEL && (y <= 0 || y >= g.h - 1);     // uniform
            const bool exact = SEL && !(TABLES && big) && !(inner_x && y > 0 && y < g.h - 1);
            float col[6];
    #pragma unroll
            for (int k = 0; k < 6; k++) col[k] = v[0][k] + v[1][k] + v[2][k];
            u32 nib = 0;
    #pragma unroll
            for (int q = 0; q < 4; q++) {
                // 3x3 neighbourhood of pixel q: columns q, q+1, q+2 of v
                const float sa[4] = {col[q],                       // left      src/stereo.c:16-28
                                     s3[0][q],                     // top       src/stereo.c:30-42
                                     p[0][q] + v[1][q],            // up-left   src/stereo.c:44-56
                                     p[2][q] + v[1][q]};           // down-left src/stereo.c:58-70
                const float sb[4] = {col[q + 2],                   // right
                                     s3[2][q],                     // bottom
                                     p[2][q + 1] + v[1][q + 2],    // down-right
                                     p[0][q + 1] + v[1][q + 2]};   // up-right
                const bool known = SEL && TABLES && big && (row_border || x + q <= 0 || x + q >= g.w - 1);
                nib |= edge_decide<TABLES>(sa, sb, tab, threshold, neg_t, exact, known) << q;
            }
            if (GHOST ? (SEL && !(in_ext && quad_in && in_y)) : !in_ext) nib = 0;
            const bool row_ok = (GHOST && !SEL) || ye < g.ext_rows;   // uniform; the last strip may be short
            if (edges != nullptr && ((GHOST && !SEL) || (quad_in && in_y && row_ok))) {
                // u8 {0,1} per pixel: bit q of the nibble -> byte q
                const u32 bytes = __umul24(nib, 0x204081u) & 0x01010101u;
                *reinterpret_cast<u32 *>(edges + img + (size_t)y * g.w + x) = bytes;
            }
            // 8 lanes x 4 bits -> one ext word, OR-reduced within each group of 8 lanes
            u32 wv = nib << (4 * (lane & 7));
            wv |= (u32)__builtin_amdgcn_update_dpp(0, (int)wv, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
            wv |= (u32)__builtin_amdgcn_update_dpp(0, (int)wv, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
            wv |= (u32)__builtin_amdgcn_update_dpp(0, (int)wv, 0x141, 0xf, 0xf, true);   // row_half_mirror
            if ((lane & 7) == 0 && wd < g.ext_words && row_ok) ext_img[(size_t)ye * g.ext_words + wd] = wv;
    #pragma unroll
            for (int k = 0; k < 6; k++) { v[0][k] = v[1][k]; v[1][k] = v[2][k]; }
    #pragma unroll
            for (int k = 0; k < 5; k++) { p[0][k] = p[1][k]; p[1][k] = p[2][k]; }
    #pragma unroll
            for (int k = 0; k < 4; k++) { s3[0][k] = s3[1][k]; s3[1][k] = s3[2][k]; }
        }
    };
    if (GHOST) {
        const int x0 = xe - 4 * lane - g.pad_l, y0 = ye0 - g.half;        // the wave's first pixel / row
        const bool inside = x0 >= 1 && x0 + 256 <= g.w - 1 && y0 >= 1 && y0 + R - 1 <= g.h - 2 &&
                            ye0 + R <= g.ext_rows && g.w >= 2 && g.h >= 2;
        if (inside) body(std::false_type{});
        else        body(std::true_type{});
    } else {
        body(std::false_type{});
    }
}

// u8 {0,1} edge image -> packed ext image (the sm_load_edges entry).  One lane
// per ext pixel; a wave's 64 values become two ext words via ballot.
__global__ __launch_bounds__(256) void k_pack_ext(const u8 *__restrict__ src_l,
                                                  const u8 *__restrict__ src_r,
                                                  u32 *__restrict__ ext, const MatchGeom g, int ghost)
{
    const int xe = blockIdx.x * blockDim.x + threadIdx.x;
    const int ye = blockIdx.y;
    const int pair = blockIdx.z >> 1, side = blockIdx.z & 1;
    const u8 *src = (side ? src_r : src_l) + (size_t)pair * g.w * g.h;
    const int x = xe - g.pad_l, y = ye - g.half;
    const bool inside = x >= 0 && x < g.w && y >= 0 && y < g.h;
    u32 val = 0;
    if (xe < g.ext_words * 32 && (inside || !ghost)) {
        const int xs = inside ? x : pos_mod(x, g.w);
        const int ys = inside ? y : pos_mod(y, g.h);
        val = src[(size_t)ys * g.w + xs] != 0;
    }
    const unsigned long long bal = __ballot(val != 0);
    if ((threadIdx.x & 63) == 0) {
        u32 *row = ext + ((size_t)blockIdx.z * g.ext_rows + ye) * g.ext_words;
        const int wd = xe >> 5;
        if (wd < g.ext_words) row[wd] = (u32)bal;
        if (wd + 1 < g.ext_words) row[wd + 1] = (u32)(bal >> 32);
    }
}

// ---------------------------------------------------------------------------
// debug tap: the per-shift planes of the reference's debug build
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_debug_planes(const u32 *__restrict__ ext, int pair,
                                                      int shift, u8 *__restrict__ match,
                                                      i32 *__restrict__ score_all,
                                                      i32 *__restrict__ scores, const MatchGeom g,
                                                      int ghost)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= g.w) return;
    const u32 *ext_l = ext + (size_t)pair * 2 * g.ext_image_words;
    const u32 *ext_r = ext_l + g.ext_image_words;
    auto bit = [&](const u32 *im, int xx, int yy) -> u32 {
        const int b = xx + g.pad_l;
        return (im[(size_t)(yy + g.half) * g.ext_words + (b >> 5)] >> (b & 31)) & 1u;
    };
    int xa = x - g.half, xb = x + g.half, ya = y - g.half, yb = y + g.half;
    if (ghost) {
        xa = max(xa, 0); xb = min(xb, g.w - 1);
        ya = max(ya, 0); yb = min(yb, g.h - 1);
    }
    i32 sum = 0;
    for (int yy = ya; yy <= yb; yy++)
        for (int xx = xa; xx <= xb; xx++)
            sum += bit(ext_l, xx, yy) == bit(ext_r, xx + shift, yy);
    const u32 m = bit(ext_l, x, y) == bit(ext_r, x + shift, y);
    const size_t o = (size_t)y * g.w + x;
    if (match) match[o] = (u8)m;
    if (score_all) score_all[o] = sum;
    if (scores) scores[o] = m ? sum : 0;
}

// ---------------------------------------------------------------------------
// step 3
// ---------------------------------------------------------------------------

// one Jacobi sweep of src/stereo.cu:235-245: read `oth`, write `cur` where
// oth == 0.  Neighbours at flat offsets +-1, +-w (the reference's unwrapped
// IDX); offsets that leave the image array are undefined in the reference and
// read as 0 here.
__global__ __launch_bounds__(256) void k_fill_holes_step(i32 *__restrict__ cur,
                                                         const i32 *__restrict__ oth, int w,
                                                         long long n)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t base = (size_t)blockIdx.y * n;
    if (oth[base + p] == 0) {
        const i32 r = p + 1 < n ? oth[base + p + 1] : 0;
        const i32 u = p + w < n ? oth[base + p + w] : 0;
        const i32 l = p - 1 >= 0 ? oth[base + p - 1] : 0;
        const i32 d = p - w >= 0 ? oth[base + p - w] : 0;
        cur[base + p] = (r + u + l + d) / 4;
    }
}

__global__ void k_count_zeros(const i32 *__restrict__ a, long long total, i32 *__restrict__ flag)
{
    bool z = false;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x)
        z |= a[p] == 0;
    if (__ballot(z) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// min / max of each pair's image AND "some pixel is 0" (the hole-fill stage only ever
// changes pixels that are 0), in one pass over the maps.  HBM-bound: 16-byte loads where
// the image allows, one atomic pair per WORKGROUP (atomics on one address serialise at
// ~12 ns each: one pair per wave of a 2048-block grid cost 190 us at 4K)
__global__ __launch_bounds__(256) void k_minmax_zero(const i32 *__restrict__ a, long long n,
                                                     i32 *__restrict__ mm, i32 *__restrict__ zero_flag)
{
    __shared__ i32 s_lo[4], s_hi[4], s_z[4];
    const i32 *img = a + (size_t)blockIdx.y * n;
    i32 lo = INT_MAX, hi = INT_MIN;
    bool z = false;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long done = 0;
    if ((((uintptr_t)img) & 15) == 0) {
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const v4i_ *q = reinterpret_cast<const v4i_ *>(img);
        const long long nq = n >> 2;
        for (long long p = tid; p < nq; p += stride) {
            const v4i_ v = q[p];
            lo = min(min(lo, v.x), min(v.y, min(v.z, v.w)));
            hi = max(max(hi, v.x), max(v.y, max(v.z, v.w)));
            z |= v.x == 0 || v.y == 0 || v.z == 0 || v.w == 0;
        }
        done = nq << 2;
    }
    for (long long p = done + tid; p < n; p += stride) {
        const i32 v = img[p];
        lo = min(lo, v);
        hi = max(hi, v);
        z |= v == 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    const bool any_z = __ballot(z) != 0;
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_z[wv] = any_z; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        bool zz = false;
        for (int k = 0; k < nw; k++) { lo = min(lo, s_lo[k]); hi = max(hi, s_hi[k]); zz |= s_z[k] != 0; }
        atomicMin(&mm[2 * blockIdx.y], lo);
        atomicMax(&mm[2 * blockIdx.y + 1], hi);
        if (zz && zero_flag) atomicOr(zero_flag, 1);
    }
}

// {INT_MAX, INT_MIN} per pair, and the "has a zero pixel" flag cleared
__global__ void k_step3_init(i32 *__restrict__ mm, int pairs, i32 *__restrict__ zero_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pairs) { mm[2 * i] = INT_MAX; mm[2 * i + 1] = INT_MIN; }
    if (i == 0 && zero_flag) *zero_flag = 0;
}

// hand the plan's flags to the host: one lane copies them into pinned host memory the
// host reads after synchronising the stream (no 4-byte hipMemcpy to pageable memory,
// which costs tens of microseconds), and clears the ones in `clear_mask`
__global__ void k_publish_flags(i32 *__restrict__ d_flags, i32 *__restrict__ h_flags, int clear_mask)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 4; i++) {
            h_flags[i] = d_flags[i];
            if ((clear_mask >> i) & 1) d_flags[i] = 0;
        }
    }
}

// src/stereo.cu:261-274
__global__ __launch_bounds__(256) void k_contour(const i32 *__restrict__ web,
                                                 const i32 *__restrict__ mm, int lines,
                                                 long long n, u8 *__restrict__ out,
                                                 i32 *__restrict__ flags)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t base = (size_t)blockIdx.y * n;
    const i32 lo = mm[2 * blockIdx.y], hi = mm[2 * blockIdx.y + 1];
    const i32 interval = lines != 0 ? (hi - lo) / lines : 0;
    if (interval == 0) {
        if (p == 0) atomicOr(&flags[0], 1);
        out[base + p] = 0;
        return;
    }
    out[base + p] = (u8)(((web[base + p] - lo) % interval) == 0);
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

static void free_timing(sm_plan *plan);
static int run_sweeps(sm_plan *plan, i32 *d_web, i32 *d_tmp, int times, int pairs, int *result_in_tmp,
                      hipStream_t st);

static int use_device(int device)
{
    SM_HIP(hipSetDevice(device));
    return SM_OK;
}
#define SM_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" int sm_device_count(int *count)
{
    if (!count) return sm_fail(SM_ERR_ARG, "sm_device_count: count is NULL");
    SM_HIP(hipGetDeviceCount(count));
    return SM_OK;
}

extern "C" int sm_malloc(int device, size_t bytes, void **d_ptr)
{
    if (!d_ptr) return sm_fail(SM_ERR_ARG, "sm_malloc: d_ptr is NULL");
    SM_TRY(use_device(device));
    *d_ptr = nullptr;
    SM_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    SM_HIP(hipMemset(*d_ptr, 0, bytes ? bytes : 1));
    return SM_OK;
}

extern "C" int sm_free(int device, void *d_ptr)
{
    SM_TRY(use_device(device));
    SM_HIP(hipFree(d_ptr));
    return SM_OK;
}

extern "C" int sm_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes)
{
    SM_TRY(use_device(device));
    SM_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return SM_OK;
}

extern "C" int sm_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes)
{
    SM_TRY(use_device(device));
    SM_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SM_OK;
}

extern "C" int sm_stream_sync(int device, void *stream)
{
    SM_TRY(use_device(device));
    SM_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SM_OK;
}

extern "C" int sm_host_alloc(size_t bytes, void **h_ptr)
{
    if (!h_ptr) return sm_fail(SM_ERR_ARG, "sm_host_alloc: h_ptr is NULL");
    *h_ptr = nullptr;
    SM_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SM_OK;
}

extern "C" int sm_host_free(void *h_ptr)
{
    SM_HIP(hipHostFree(h_ptr));
    return SM_OK;
}

extern "C" int sm_memcpy_h2d_async(int device, void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    SM_TRY(use_device(device));
    SM_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SM_OK;
}

extern "C" int sm_memcpy_d2h_async(int device, void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    SM_TRY(use_device(device));
    SM_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SM_OK;
}

extern "C" int sm_stream_create(int device, void **stream)
{
    if (!stream) return sm_fail(SM_ERR_ARG, "sm_stream_create: stream is NULL");
    SM_TRY(use_device(device));
    hipStream_t st = nullptr;
    SM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return SM_OK;
}

extern "C" int sm_stream_destroy(int device, void *stream)
{
    SM_TRY(use_device(device));
    SM_HIP(hipStreamDestroy((hipStream_t)stream));
    return SM_OK;
}

extern "C" int sm_event_create(int device, void **event)
{
    if (!event) return sm_fail(SM_ERR_ARG, "sm_event_create: event is NULL");
    SM_TRY(use_device(device));
    hipEvent_t ev = nullptr;
    SM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *event = (void *)ev;
    return SM_OK;
}

extern "C" int sm_event_destroy(int device, void *event)
{
    SM_TRY(use_device(device));
    SM_HIP(hipEventDestroy((hipEvent_t)event));
    return SM_OK;
}

extern "C" int sm_event_record(int device, void *event, void *stream)
{
    if (!event) return sm_fail(SM_ERR_ARG, "sm_event_record: event is NULL");
    SM_TRY(use_device(device));
    SM_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SM_OK;
}

extern "C" int sm_stream_wait_event(int device, void *stream, void *event)
{
    if (!event) return sm_fail(SM_ERR_ARG, "sm_stream_wait_event: event is NULL");
    SM_TRY(use_device(device));
    SM_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return SM_OK;
}

extern "C" int sm_event_sync(int device, void *event)
{
    if (!event) return sm_fail(SM_ERR_ARG, "sm_event_sync: event is NULL");
    SM_TRY(use_device(device));
    SM_HIP(hipEventSynchronize((hipEvent_t)event));
    return SM_OK;
}

extern "C" int sm_plan_create(int device, int width, int height, int num_shifts,
                              int square_width, int border, int max_pairs, sm_plan **out)
{
    return sm_plan_create_ex(device, width, height, num_shifts, square_width, border, max_pairs, nullptr, out);
}

extern "C" int sm_plan_create_ex(int device, int width, int height, int num_shifts, int square_width,
                                 int border, int max_pairs, const sm_plan_options *options, sm_plan **out)
{
    if (!out) return sm_fail(SM_ERR_ARG, "sm_plan_create: out is NULL");
    // struct_size: 0 = "nothing but the size field" (an all-zero struct is the plan's own choice throughout);
    // a LARGER struct comes from a caller built against a newer header: the prefix this library knows is taken
    if (options && (options->struct_size < 0 || (options->struct_size > 0 && options->struct_size < (int)sizeof(int))))
        return sm_fail(SM_ERR_ARG, "sm_plan_create_ex: options->struct_size %d is not that of a sm_plan_options "
                       "(this library: %d bytes)", options->struct_size, (int)sizeof(sm_plan_options));
    *out = nullptr;
    if (width < 1 || height < 1)
        return sm_fail(SM_ERR_ARG, "sm_plan_create: image size %dx%d is not positive", width, height);
    if ((long long)width * height > (1ll << 30))
        return sm_fail(SM_ERR_ARG, "sm_plan_create: image of %dx%d pixels is too large", width, height);
    if (num_shifts < 1 || num_shifts > 65535)
        return sm_fail(SM_ERR_ARG, "sm_plan_create: num_shifts %d outside 1..65535", num_shifts);
    if (square_width < 0)
        return sm_fail(SM_ERR_ARG, "sm_plan_create: square_width %d is negative "
                       "(undefined in the reference)", square_width);
    if (square_width > width || square_width > height)
        return sm_fail(SM_ERR_ARG, "error: square width must not be higher than image width/height");
    if (border != SM_TOROIDAL && border != SM_GHOST)
        return sm_fail(SM_ERR_ARG, "sm_plan_create: border %d is neither SM_TOROIDAL nor SM_GHOST", border);
    if (max_pairs < 1)
        return sm_fail(SM_ERR_ARG, "sm_plan_create: max_pairs %d < 1", max_pairs);
    SM_TRY(use_device(device));

    sm_plan *p = (sm_plan *)calloc(1, sizeof *p);
    if (p) p->timing_every = 1;
    if (!p) return sm_fail(SM_ERR_NOMEM, "error: out of memory");
    p->device = device;
    p->width = width; p->height = height;
    p->num_shifts = num_shifts; p->square_width = square_width;
    p->border = border; p->max_pairs = max_pairs;
    if (options && options->struct_size > 0)      // a shorter (older) struct: the rest stays 0; a longer one: the known prefix
        memcpy(&p->opt, options, std::min((size_t)options->struct_size, sizeof(sm_plan_options)));
    p->opt.struct_size = (int)sizeof(sm_plan_options);
    int rc = sm_match_configure(p);
    if (rc) { free(p); return rc; }
    p->g.web_bytes = 4;

    p->ext_bytes = (size_t)max_pairs * 2 * (size_t)p->g.ext_image_words * sizeof(u32);
    hipError_t e = hipMalloc((void **)&p->d_ext_buf[0], p->ext_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_ext_buf[1], p->ext_bytes);
    if (e == hipSuccess) e = hipMemset(p->d_ext_buf[1], 0, p->ext_bytes);
    for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipStreamCreateWithFlags(&p->lane[b], hipStreamNonBlocking);
    for (int q = 0; q < 4 && e == hipSuccess; q++) e = hipEventCreateWithFlags(&p->ev_free[q], hipEventDisableTiming);
    p->d_ext = p->d_ext_buf[0];
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_inputs, hipEventDisableTiming);
    for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipEventCreateWithFlags(&p->ev_fork[b], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_flags, 4 * sizeof(i32));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_edge_tab, 768 * sizeof(u32));
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->h_flags, 4 * sizeof(i32), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(p->d_ext, 0, p->ext_bytes);
    if (e == hipSuccess) e = hipMemset(p->d_flags, 0, 4 * sizeof(i32));
    // (kernels without a narrow store path write narrow maps through an int32 staging map,
    // max_pairs * W * H * 4 bytes: NOT allocated here -- a plan whose caller only ever asks for
    // int32 maps must not pay for it -- but by sm_plan_reserve_narrow, or by the first narrow request)
    if (e != hipSuccess) {
        for (int b = 0; b < 2; b++) {
            if (p->d_ext_buf[b]) (void)hipFree(p->d_ext_buf[b]);
            if (p->lane[b]) (void)hipStreamDestroy(p->lane[b]);
        }
        for (int q = 0; q < 4; q++)
            if (p->ev_free[q]) (void)hipEventDestroy(p->ev_free[q]);
        if (p->ev_inputs) (void)hipEventDestroy(p->ev_inputs);
        for (int b = 0; b < 2; b++)
            if (p->ev_fork[b]) (void)hipEventDestroy(p->ev_fork[b]);
        if (p->d_flags) (void)hipFree(p->d_flags);
        if (p->d_edge_tab) (void)hipFree(p->d_edge_tab);
        if (p->d_web_tmp) (void)hipFree(p->d_web_tmp);
        if (p->h_flags) (void)hipHostFree(p->h_flags);
        free(p);
        return sm_fail(e == hipErrorOutOfMemory ? SM_ERR_NOMEM : SM_ERR_HIP,
                       "sm_plan_create: workspace allocation failed: %s", hipGetErrorString(e));
    }
    // Resolve the code objects of the kernels this plan will launch now (setup),
    // so that the first timed launch does not pay the runtime's lazy loading.
    {
        hipFuncAttributes fa;
        const bool gh = border == SM_GHOST;
        const void *fns[] = {
            (const void *)k_edge_thresholds, (const void *)k_pack_ext, (const void *)k_debug_planes,
            (const void *)k_fill_holes_step, (const void *)k_count_zeros, (const void *)k_step3_init,
            (const void *)k_contour, (const void *)k_minmax_zero, (const void *)k_publish_flags,
            gh ? (const void *)k_edges_ext4<true, true> : (const void *)k_edges_ext4<false, true>,
            gh ? (const void *)k_edges_ext4<true, true, true> : (const void *)k_edges_ext4<false, true, true>,
            gh ? (const void *)k_edges_ext<true, true> : (const void *)k_edges_ext<false, true>,
        };
        for (const void *f : fns) (void)hipFuncGetAttributes(&fa, f);
    }
    if (p->kernel == SM_KERNEL_BS) {
        rc = sm_bs_prepare(p);
        if (rc) { sm_plan_destroy(p); return rc; }
    }
    *out = p;
    return SM_OK;
}

// the int32 staging map of narrow results for the kernels that have no narrow store path
static int reserve_narrow(sm_plan *plan, const char *me)
{
    if (plan->d_web_tmp || plan->kernel == SM_KERNEL_BS) return SM_OK;
    const size_t bytes = (size_t)plan->max_pairs * plan->width * plan->height * sizeof(i32);
    const hipError_t e = hipMalloc((void **)&plan->d_web_tmp, bytes);
    if (e != hipSuccess) {
        plan->d_web_tmp = nullptr;
        return sm_fail(e == hipErrorOutOfMemory ? SM_ERR_NOMEM : SM_ERR_HIP,
                       "%s: %zu bytes for the int32 staging map of narrow results: %s", me, bytes, hipGetErrorString(e));
    }
    return SM_OK;
}

extern "C" int sm_plan_reserve_narrow(sm_plan *plan)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_plan_reserve_narrow: plan is NULL");
    SM_TRY(use_device(plan->device));
    return reserve_narrow(plan, "sm_plan_reserve_narrow");
}

extern "C" void sm_plan_destroy(sm_plan *plan)
{
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    for (int b = 0; b < 2; b++) (void)hipStreamSynchronize(plan->lane[b]);
    free_timing(plan);
    for (int b = 0; b < 2; b++) {
        (void)hipFree(plan->d_ext_buf[b]);
        (void)hipStreamDestroy(plan->lane[b]);
    }
    for (int q = 0; q < 4; q++) (void)hipEventDestroy(plan->ev_free[q]);
    (void)hipEventDestroy(plan->ev_inputs);
    for (int b = 0; b < 2; b++) (void)hipEventDestroy(plan->ev_fork[b]);
    if (plan->d_web_tmp) (void)hipFree(plan->d_web_tmp);
    (void)hipFree(plan->d_flags);
    (void)hipFree(plan->d_edge_tab);
    (void)hipHostFree(plan->h_flags);
    free(plan);
}

extern "C" const char *sm_plan_describe(const sm_plan *plan) { return plan ? plan->describe : ""; }

extern "C" int sm_plan_geometry_sized(const sm_plan *plan, sm_geometry *out_any, size_t size)
{
    if (!plan || !out_any) return sm_fail(SM_ERR_ARG, "sm_plan_geometry: NULL argument");
    if (size < sizeof(int)) return sm_fail(SM_ERR_ARG, "sm_plan_geometry_sized: size %zu is not that of a sm_geometry", size);
    sm_geometry full, *out = &full;
    memset(&full, 0, sizeof full);
    const MatchGeom &g = plan->g;
    out->kernel = plan->kernel;
    out->window = g.n;
    out->shifts_per_lane = g.ds;
    out->shift_lanes = g.nl;
    out->threads = g.threads;
    out->tile_w = g.tw;
    out->tile_h = g.duo ? 2 * g.tile_h : g.tile_h;
    out->tiles_x = g.tiles_x;
    out->tiles_y = g.tiles_y;
    out->ext_words = g.ext_words;
    out->ext_rows = g.ext_rows;
    out->pad_l = g.pad_l;
    out->lds_bytes = g.lds_bytes;
    out->two_wave_variant = g.cap2;
    out->edge_rows_per_wave = (g.w % 4 == 0) ? SM_EDGE4_ROWS : SM_EDGE_ROWS;
    out->waves_per_workgroup = plan->kernel == SM_KERNEL_BS ? (g.duo ? 2 : 1) : (g.threads + 63) / 64;
    out->lane_merge_lds = plan->kernel == SM_KERNEL_BS && g.xmerge;
    // the caller's struct may be older (shorter: it gets the fields it knows) or newer (longer: the rest is zeroed)
    memset(out_any, 0, size);
    memcpy(out_any, &full, size < sizeof full ? size : sizeof full);
    return SM_OK;
}

extern "C" int sm_plan_geometry(const sm_plan *plan, sm_geometry *out)
{
    return sm_plan_geometry_sized(plan, out, sizeof(sm_geometry));
}

extern "C" size_t sm_plan_workspace_bytes(const sm_plan *plan)
{
    if (!plan) return 0;
    const size_t staging = plan->d_web_tmp ? (size_t)plan->max_pairs * plan->width * plan->height * sizeof(i32) : 0;
    return 2 * plan->ext_bytes + 4 * sizeof(i32) + 768 * sizeof(u32) + staging;
}

// synchronise `st` and return the plan's flags as they were at that point; flags in
// clear_mask are reset on the device
static int read_flags(sm_plan *plan, hipStream_t st, int clear_mask, i32 out[4])
{
    hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, st, plan->d_flags, plan->h_flags, clear_mask);
    SM_LAUNCH_CHECK("k_publish_flags");
    SM_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 4; i++) out[i] = plan->h_flags[i];
    return SM_OK;
}

// Is `st` recording into a graph (hipStreamBeginCapture, torch.cuda.graph)?  Asked only on the paths that cannot be
// captured or need another protocol inside a capture: the steady state of a plain plan never calls it.
static bool stream_capturing(hipStream_t st, unsigned long long *id = nullptr)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    if (hipStreamGetCaptureInfo(st, &cs, &cid) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (id) *id = cid;
    return cs != hipStreamCaptureStatusNone;
}

static int check_plan_pairs(const sm_plan *plan, int pairs, const char *who)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "%s: plan is NULL", who);
    if (pairs < 1 || pairs > plan->max_pairs)
        return sm_fail(SM_ERR_ARG, "%s: pairs %d outside 1..%d (max_pairs of the plan)", who, pairs,
                       plan->max_pairs);
    return SM_OK;
}

static int pack_ext(sm_plan *plan, const u8 *l, const u8 *r, int pairs, hipStream_t st)
{
    const MatchGeom &g = plan->g;
    const dim3 grid((g.ext_words * 32 + 255) / 256, g.ext_rows, pairs * 2), block(256);
    hipLaunchKernelGGL(k_pack_ext, grid, block, 0, st, l, r, plan->d_ext, g,
                       plan->border == SM_GHOST ? 1 : 0);
    SM_LAUNCH_CHECK("k_pack_ext");
    plan->pairs_loaded = pairs;
    return SM_OK;
}

static float edge_neg_t(double threshold)
{
    return -(float)(threshold * 0.5);
}

// decision tables depend on the threshold only: rebuilt when it changes
static int ensure_edge_tables(sm_plan *plan, double threshold, hipStream_t st)
{
    if (plan->tab_valid && memcmp(&plan->tab_threshold, &threshold, sizeof threshold) == 0)
        return SM_OK;
    if (stream_capturing(st))
        return sm_fail(SM_ERR_ARG, "the decision tables of threshold %g are not prepared and the stream is capturing: their "
                       "set-up reads a verdict back to the host, which a graph cannot hold -- call "
                       "sm_plan_prepare_threshold(plan, threshold, stream) before the capture begins", threshold);
    SM_HIP(hipMemsetAsync(&plan->d_flags[2], 0, sizeof(i32), st));
    hipLaunchKernelGGL(k_edge_thresholds, dim3(766), dim3(256), 0, st, threshold,
                       plan->d_edge_tab, &plan->d_flags[2]);
    SM_LAUNCH_CHECK("k_edge_thresholds");
    // read the verdict back once per new threshold (not in the steady state): it
    // selects the kernel instantiation
    i32 f[4];
    SM_TRY(read_flags(plan, st, 0, f));
    plan->tab_ok = f[2] == 0;
    plan->tab_threshold = threshold;
    plan->tab_valid = 1;
    return SM_OK;
}

extern "C" int sm_debug_edge_table_fast(sm_plan *plan, double threshold, uint8_t *d_table,
                                        int *not_threshold_form, void *stream)
{
    if (!plan || !d_table || !not_threshold_form)
        return sm_fail(SM_ERR_ARG, "sm_debug_edge_table_fast: NULL argument");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    SM_TRY(ensure_edge_tables(plan, threshold, st));
    hipLaunchKernelGGL(k_edge_table_fast, dim3(3, 766), dim3(256), 0, st, plan->d_edge_tab,
                       edge_neg_t(threshold), d_table);
    SM_LAUNCH_CHECK("k_edge_table_fast");
    i32 f[4];
    SM_TRY(read_flags(plan, st, 0, f));
    *not_threshold_form = f[2];
    return SM_OK;
}

extern "C" int sm_plan_prepare_threshold(sm_plan *plan, double threshold, void *stream)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_plan_prepare_threshold: plan is NULL");
    if (!(threshold >= 0.0 && threshold <= 1.0))
        return sm_fail(SM_ERR_ARG, "error: threshold must be between 0 and 1");
    SM_TRY(use_device(plan->device));
    return ensure_edge_tables(plan, threshold, (hipStream_t)stream);
}

extern "C" int sm_find_edges(sm_plan *plan, const uint8_t *d_gray_left,
                             const uint8_t *d_gray_right, double threshold, int pairs,
                             uint8_t *d_edges_left, uint8_t *d_edges_right, void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_find_edges"));
    if (!d_gray_left || !d_gray_right)
        return sm_fail(SM_ERR_ARG, "sm_find_edges: input image pointer is NULL");
    if (!(threshold >= 0.0 && threshold <= 1.0))
        return sm_fail(SM_ERR_ARG, "error: threshold must be between 0 and 1");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    SM_TRY(ensure_edge_tables(plan, threshold, st));
    const MatchGeom &g = plan->g;
    const dim3 grid((g.edge_words_r * 32 + 255) / 256, (g.ext_rows + SM_EDGE_ROWS - 1) / SM_EDGE_ROWS,
                    pairs * 2), block(256);
    const bool ghost = plan->border == SM_GHOST;
    // the 4-pixels-per-lane kernel moves dwords: rows (w % 4 == 0) and base pointers
    // must be 4-byte aligned, else the any-width kernel takes over
    const bool aligned4 = (((uintptr_t)d_gray_left | (uintptr_t)d_gray_right |
                            (uintptr_t)d_edges_left | (uintptr_t)d_edges_right) & 3) == 0;
    if (g.w % 4 == 0 && aligned4 && plan->opt.edge_kernel != 1) {
        const int strips = (g.ext_rows + SM_EDGE4_ROWS - 1) / SM_EDGE4_ROWS;
        const int lanes = g.edge_words_r * 8;       // (the left image's waves beyond its own need leave at once)
        // waves side by side, unless that rounds the row up by more than 3 % (see the kernel)
        const bool stacked = (lanes + 255) / 256 * 256 > lanes + lanes / 32;
        const dim3 grid4 = stacked ? dim3((lanes + 63) / 64, (strips + 3) / 4, pairs * 2)
                                   : dim3((lanes + 255) / 256, strips, pairs * 2);
#define SM_EDGES_GO(G, T)                                                                      \
    do {                                                                                       \
        if (stacked)                                                                           \
            hipLaunchKernelGGL((k_edges_ext4<G, T, true>), grid4, block, 0, st, d_gray_left, d_gray_right, \
                               d_edges_left, d_edges_right, plan->d_ext, plan->d_edge_tab, g, threshold,   \
                               edge_neg_t(threshold));                                         \
        else                                                                                   \
            hipLaunchKernelGGL((k_edges_ext4<G, T, false>), grid4, block, 0, st, d_gray_left, d_gray_right, \
                               d_edges_left, d_edges_right, plan->d_ext, plan->d_edge_tab, g, threshold,   \
                               edge_neg_t(threshold));                                         \
    } while (0)
        if (plan->tab_ok) { if (ghost) SM_EDGES_GO(true, true); else SM_EDGES_GO(false, true); }
        else              { if (ghost) SM_EDGES_GO(true, false); else SM_EDGES_GO(false, false); }
#undef SM_EDGES_GO
    } else {
#define SM_EDGES_GO(G, T)                                                                      \
    hipLaunchKernelGGL((k_edges_ext<G, T>), grid, block, 0, st, d_gray_left, d_gray_right,       \
                       d_edges_left, d_edges_right, plan->d_ext, plan->d_edge_tab, g, threshold, \
                       edge_neg_t(threshold))
        if (plan->tab_ok) { if (ghost) SM_EDGES_GO(true, true); else SM_EDGES_GO(false, true); }
        else              { if (ghost) SM_EDGES_GO(true, false); else SM_EDGES_GO(false, false); }
#undef SM_EDGES_GO
    }
    SM_LAUNCH_CHECK("k_edges_ext");
    plan->pairs_loaded = pairs;
    return SM_OK;
}

extern "C" int sm_load_edges(sm_plan *plan, const uint8_t *d_edges_left,
                             const uint8_t *d_edges_right, int pairs, void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_load_edges"));
    if (!d_edges_left || !d_edges_right)
        return sm_fail(SM_ERR_ARG, "sm_load_edges: edge image pointer is NULL");
    SM_TRY(use_device(plan->device));
    return pack_ext(plan, d_edges_left, d_edges_right, pairs, (hipStream_t)stream);
}

// int32 web -> uint16 / uint8 (the kernels that have no narrow store path of their own)
__global__ __launch_bounds__(256) void k_narrow_web(const i32 *__restrict__ src, void *__restrict__ dst,
                                                   long long n, int bytes)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (bytes == 1) ((u8 *)dst)[p] = (u8)src[p];
    else ((unsigned short *)dst)[p] = (unsigned short)src[p];
}

extern "C" int sm_match_wta(sm_plan *plan, int pairs, int32_t *d_web, int32_t *d_best,
                            void *stream)
{
    return sm_match_wta_typed(plan, pairs, d_web, SM_WEB_I32, d_best, stream);
}

extern "C" int sm_match_wta_typed(sm_plan *plan, int pairs, void *d_web_any, int web_type,
                                  int32_t *d_best, void *stream)
{
    const char *me = web_type == SM_WEB_I32 ? "sm_match_wta" : "sm_match_wta_typed";
    SM_TRY(check_plan_pairs(plan, pairs, me));
    if (!d_web_any) return sm_fail(SM_ERR_ARG, "%s: d_web is NULL", me);
    if (web_type != SM_WEB_I32 && web_type != SM_WEB_U16 && web_type != SM_WEB_U8)
        return sm_fail(SM_ERR_ARG, "%s: web_type %d is not SM_WEB_I32/U16/U8", me, web_type);
    if ((web_type == SM_WEB_U8 && plan->num_shifts > 255) || (web_type == SM_WEB_U16 && plan->num_shifts > 65535))
        return sm_fail(SM_ERR_ARG, "%s: %d shifts do not fit the requested web type", me, plan->num_shifts);
    const int web_bytes = web_type == SM_WEB_I32 ? 4 : web_type == SM_WEB_U16 ? 2 : 1;
    int32_t *d_web = (int32_t *)d_web_any;
    // kernels without a narrow store path: int32 into the plan's staging map (allocated with the
    // plan), then narrow.  ONE staging map per plan: see the threading note in stereo_hip.h
    const bool via_tmp = web_bytes != 4 && plan->kernel != SM_KERNEL_BS;
    if (via_tmp) {
        if (!plan->d_web_tmp) {         // the first narrow request on such a plan (sm_plan_reserve_narrow keeps
            SM_TRY(use_device(plan->device));      // this allocation, which synchronises the device, out of a timed path)
            if (stream_capturing((hipStream_t)stream))
                return sm_fail(SM_ERR_ARG, "%s: the int32 staging map of narrow results is not allocated and the stream is "
                               "capturing (an allocation cannot be captured): call sm_plan_reserve_narrow(plan) first", me);
            SM_TRY(reserve_narrow(plan, me));
        }
        d_web = plan->d_web_tmp;
    }
    if (pairs > plan->pairs_loaded)
        return sm_fail(SM_ERR_ARG, "%s: %d pairs requested but edges of only %d are loaded "
                       "(call sm_find_edges or sm_load_edges first)", me, pairs, plan->pairs_loaded);
    SM_TRY(use_device(plan->device));
    // event records are not free (~4 us each on the launch stream): time a sample of
    // the launches, and record the buffer-release event only when someone can wait on it
    const bool timed = plan->timing_n < plan->timing_cap &&
                       plan->timing_seen++ % plan->timing_every == 0;
    if (timed && stream_capturing((hipStream_t)stream))
        return sm_fail(SM_ERR_ARG, "%s: kernel timing is armed (sm_plan_time_kernels) and the stream is capturing: the "
                       "timing events of a launch cannot be read back from a graph -- disarm with "
                       "sm_plan_time_kernels(plan, 0) before the capture begins", me);
    // The bit-sliced kernel's launcher attaches the two events to the dispatch packet itself
    // (the completion signal's own start / end time stamps): no extra packets on the stream.
    // Separate event records cost ~4 us each there, 6 % of a 4K step when every second launch
    // is timed (bench.py at --steps 20).  Other kernels keep the bracketing records.
    const bool attach = timed && plan->kernel == SM_KERNEL_BS && !via_tmp && !plan->opt.timing_by_records;
    if (timed && !attach) SM_HIP(hipEventRecord(plan->t_begin[plan->timing_n], (hipStream_t)stream));
    {
        // what this launch adds to the plan's geometry, by value (the plan itself is not touched):
        // int4 stores need 16-byte aligned maps, otherwise this launch stores scalars; the element
        // size of the web map; the events of a timed launch
        MatchLaunch l;
        l.g = plan->g;
        const int kb = via_tmp ? 4 : web_bytes;
        if (((uintptr_t)d_web & (4 * kb - 1)) != 0 || ((uintptr_t)d_best & 15) != 0) l.g.vec_ok = 0;
        l.g.web_bytes = kb;
        l.ev_begin = attach ? plan->t_begin[plan->timing_n] : nullptr;
        l.ev_end = attach ? plan->t_end[plan->timing_n] : nullptr;
        const int rc = sm_match_launch(plan, l, pairs, d_web, d_best, (hipStream_t)stream);
        if (rc) return rc;
        if (via_tmp) {
            const long long n = (long long)pairs * plan->width * plan->height;
            hipLaunchKernelGGL(k_narrow_web, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                               (hipStream_t)stream, d_web, d_web_any, n, web_bytes);
            SM_LAUNCH_CHECK("k_narrow_web");
        }
    }
    if (timed && !attach) SM_HIP(hipEventRecord(plan->t_end[plan->timing_n], (hipStream_t)stream));
    if (timed) plan->timing_n++;
    if (plan->pipelined) {
        // the release event of pipelined call number seq: `stream` of sm_run waits for it, and so do
        // later calls that must not overtake this one
        SM_HIP(hipEventRecord(plan->ev_free[plan->seq & 3], (hipStream_t)stream));
        plan->ev_free_set[plan->seq & 3] = 1;
    } else {
        plan->unfenced = 1;                // launches a later pipelined phase has no event for
    }
    return SM_OK;
}

static void free_timing(sm_plan *plan)
{
    for (int i = 0; i < plan->timing_cap; i++) {
        (void)hipEventDestroy(plan->t_begin[i]);
        (void)hipEventDestroy(plan->t_end[i]);
    }
    free(plan->t_begin);
    free(plan->t_end);
    plan->t_begin = plan->t_end = nullptr;
    plan->timing_cap = plan->timing_n = 0;
}

extern "C" int sm_plan_time_kernels(sm_plan *plan, int capacity)
{
    if (!plan || capacity < 0 || capacity > (1 << 20))
        return sm_fail(SM_ERR_ARG, "sm_plan_time_kernels: bad argument");
    SM_TRY(use_device(plan->device));
    plan->timing_seen = 0;
    if (capacity == plan->timing_cap) { plan->timing_n = 0; return SM_OK; }
    free_timing(plan);
    if (capacity == 0) return SM_OK;
    plan->t_begin = (hipEvent_t *)calloc(capacity, sizeof(hipEvent_t));
    plan->t_end = (hipEvent_t *)calloc(capacity, sizeof(hipEvent_t));
    if (!plan->t_begin || !plan->t_end) return sm_fail(SM_ERR_NOMEM, "error: out of memory");
    for (int i = 0; i < capacity; i++) {
        SM_HIP(hipEventCreate(&plan->t_begin[i]));
        SM_HIP(hipEventCreate(&plan->t_end[i]));
        plan->timing_cap = i + 1;
    }
    return SM_OK;
}

extern "C" int sm_plan_time_stride(sm_plan *plan, int every)
{
    if (!plan || every < 1) return sm_fail(SM_ERR_ARG, "sm_plan_time_stride: bad argument");
    plan->timing_every = every;
    plan->timing_seen = 0;
    return SM_OK;
}

extern "C" int sm_plan_kernel_ms(sm_plan *plan, double *mean_ms, int *launches)
{
    if (!plan || !mean_ms || !launches) return sm_fail(SM_ERR_ARG, "sm_plan_kernel_ms: NULL argument");
    SM_TRY(use_device(plan->device));
    double sum = 0;
    for (int i = 0; i < plan->timing_n; i++) {
        float ms = 0;
        SM_HIP(hipEventSynchronize(plan->t_end[i]));
        SM_HIP(hipEventElapsedTime(&ms, plan->t_begin[i], plan->t_end[i]));
        sum += ms;
    }
    *launches = plan->timing_n;
    *mean_ms = plan->timing_n ? sum / plan->timing_n : 0.0;
    return SM_OK;
}

extern "C" int sm_plan_set_pipelined(sm_plan *plan, int enabled)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_plan_set_pipelined: plan is NULL");
    plan->pipelined = enabled == 2 ? 2 : (enabled != 0);
    return SM_OK;
}

extern "C" int sm_run(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                      double threshold, int pairs, int32_t *d_web, int32_t *d_best, void *stream)
{
    return sm_run_typed(plan, d_gray_left, d_gray_right, threshold, pairs, d_web, SM_WEB_I32, d_best, stream);
}

static int run_on_lanes(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                        double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                        void *stream, hipEvent_t inputs_ready);

extern "C" int sm_run_typed(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                            double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                            void *stream)
{
    if (!plan || !plan->pipelined) {
        SM_TRY(sm_find_edges(plan, d_gray_left, d_gray_right, threshold, pairs, nullptr, nullptr, stream));
        return sm_match_wta_typed(plan, pairs, d_web, web_type, d_best, stream);
    }
    return run_on_lanes(plan, d_gray_left, d_gray_right, threshold, pairs, d_web, web_type, d_best, stream, nullptr);
}

// sm_run whose ONLY input dependency is an event (DESIGN.md 9.4 of round 4; replaces the synchronous upload in front of
// every call, src/stereo.cu:402-403): the call is free to overlap with the one before it, and the plan takes the two
// lanes by itself where that pays -- a match launch that does not fill the chip twice over (fewer than 2 x 1024 waves:
// a lone pair up to 4K), or a plan set pipelined.
extern "C" int sm_run_after(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                            double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                            void *stream, void *inputs_ready_event)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_run_after"));
    const MatchGeom &g = plan->g;
    const long long waves = (long long)g.tiles_x * g.tiles_y * pairs * ((g.threads + 63) / 64);
    // (... and no more than 128 shifts: the edge detection the overlap hides is then a ninth of a step or more.  At 256 shifts --
    // C5: a 158 us match launch beside 15 us of edges -- two calls sharing the chip cost more than that: 0.1782 against 0.1755 ms
    // per step, where C3 gains 2.6 % and C1 / C2 9-19 %: profiles/r05/bench_all_configs.txt)
    if (!plan->pipelined && (waves >= 2 * 1024 || plan->num_shifts > 128)) {
        SM_TRY(use_device(plan->device));
        if (inputs_ready_event) SM_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)inputs_ready_event, 0));
        SM_TRY(sm_find_edges(plan, d_gray_left, d_gray_right, threshold, pairs, nullptr, nullptr, stream));
        return sm_match_wta_typed(plan, pairs, d_web, web_type, d_best, stream);
    }
    const int was = plan->pipelined;
    if (!was) plan->pipelined = 1;          // (the match launch records the call's release event when the plan is pipelined)
    const int rc = run_on_lanes(plan, d_gray_left, d_gray_right, threshold, pairs, d_web, web_type, d_best, stream,
                                (hipEvent_t)inputs_ready_event);
    plan->pipelined = was;
    return rc;
}

static int run_on_lanes(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                        double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                        void *stream, hipEvent_t inputs_ready)
{
    // pipelined: call q runs on one of two lanes (internal streams, alternating): its edge detection
    // into the lane's own ext buffer, then its match launch, in stream order.  Nothing orders call q
    // against call q - 1 on the other lane, so the edges of call q run beside the match of call q - 1,
    // and the first waves of match q take the SIMD slots that the early finishers of match q - 1 leave
    // (the younger wave of every SIMD pair ends alone).  Call q - 3 (the one before q - 1
    // on the other lane) has finished before q starts: at most two calls are in flight.  `stream`
    // waits for the call's release event: work the caller puts on it afterwards sees the results.
    SM_TRY(check_plan_pairs(plan, pairs, "sm_run"));
    SM_TRY(use_device(plan->device));
    const int b = plan->cur ^ 1;
    hipStream_t lane = plan->lane[b], user = (hipStream_t)stream;
    const unsigned q = plan->seq + 1;
    // what two calls in flight could share: the threshold tables (rebuilt when the threshold
    // changes), the one int32 staging map of the kernels without a narrow store path, and result
    // maps the caller hands to consecutive calls -- any of these puts call q behind call q - 1
    const size_t px = (size_t)pairs * plan->width * plan->height;
    const uintptr_t lo[2] = {(uintptr_t)d_web, (uintptr_t)d_best};
    const uintptr_t hi[2] = {lo[0] + px * (web_type == SM_WEB_I32 ? 4 : web_type == SM_WEB_U16 ? 2 : 1),
                             d_best ? lo[1] + px * 4 : 0};
    bool shared = !(plan->tab_valid && memcmp(&plan->tab_threshold, &threshold, sizeof threshold) == 0) ||
                  (web_type != SM_WEB_I32 && plan->kernel != SM_KERNEL_BS);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            if (lo[i] < plan->out_hi[j] && plan->out_lo[j] < hi[i]) shared = true;

    unsigned long long cap_id = 0;
    if (stream_capturing(user, &cap_id)) {
        // INSIDE A STREAM CAPTURE every operation must descend from the capturing stream and join it again, and no
        // event recorded outside the capture may be waited for (hipErrorStreamCaptureIsolation -- what round 4's
        // attempt ran into: its lanes waited for the release events of calls made before the capture began;
        // tools/capture_probe.hip, profiles/r05/capture_probe.txt).  Protocol: lane b leaves `stream` at ev_fork[b],
        // which the PREVIOUS captured call recorded before it joined its own lane back -- so call q depends on
        // everything up to call q - 2 and runs beside call q - 1 in the graph, as outside a capture -- and every
        // call joins its lane back at once (`stream` waits for its release event), so the capture can end anywhere.
        // (what cannot be captured is refused BEFORE the lane leaves `stream`: an error must not leave the capture unjoined)
        if (!(plan->tab_valid && memcmp(&plan->tab_threshold, &threshold, sizeof threshold) == 0))
            return sm_fail(SM_ERR_ARG, "sm_run: the decision tables of threshold %g are not prepared and the stream is capturing: "
                           "call sm_plan_prepare_threshold(plan, threshold, stream) before the capture begins", threshold);
        if (plan->timing_n < plan->timing_cap)
            return sm_fail(SM_ERR_ARG, "sm_run: kernel timing is armed (sm_plan_time_kernels) and the stream is capturing: "
                           "disarm with sm_plan_time_kernels(plan, 0) before the capture begins");
        if (web_type != SM_WEB_I32 && plan->kernel != SM_KERNEL_BS && !plan->d_web_tmp)
            return sm_fail(SM_ERR_ARG, "sm_run: the int32 staging map of narrow results is not allocated and the stream is "
                           "capturing: call sm_plan_reserve_narrow(plan) first");
        const bool first = !plan->cap_live || plan->cap_id != cap_id;
        if (first) {
            plan->cap_live = 1;
            plan->cap_id = cap_id;
            SM_HIP(hipEventRecord(plan->ev_fork[b], user));
        }
        SM_HIP(hipStreamWaitEvent(lane, plan->ev_fork[b], 0));
        if (inputs_ready) SM_HIP(hipStreamWaitEvent(lane, inputs_ready, 0));     // (an event of this capture, or the call fails)
        if (shared && !first) SM_HIP(hipStreamWaitEvent(lane, plan->ev_free[(q - 1) & 3], 0));
        plan->seq = q;
        plan->cur = b;
        plan->d_ext = plan->d_ext_buf[b];
        for (int i = 0; i < 2; i++) { plan->out_lo[i] = lo[i]; plan->out_hi[i] = hi[i]; }
        SM_TRY(sm_find_edges(plan, d_gray_left, d_gray_right, threshold, pairs, nullptr, nullptr, (void *)lane));
        SM_TRY(sm_match_wta_typed(plan, pairs, d_web, web_type, d_best, (void *)lane));   // records ev_free[q & 3] on the lane
        SM_HIP(hipEventRecord(plan->ev_fork[b ^ 1], user));
        SM_HIP(hipStreamWaitEvent(user, plan->ev_free[q & 3], 0));
        // the events of a capture are nodes of its graph: nothing outside it waits for them, and the next call outside a
        // capture orders its lanes behind `stream` (where the graph is launched, if it is)
        for (int i = 0; i < 4; i++) plan->ev_free_set[i] = 0;
        plan->unfenced = 1;
        return SM_OK;
    }
    plan->cap_live = 0;
    if (plan->unfenced || plan->pipelined == 2) {
        // work already on `stream` that a lane must not overtake: the launches of a sequential phase
        // (once, both lanes) or, in ordered mode, whatever produces this call's inputs (this lane)
        SM_HIP(hipEventRecord(plan->ev_inputs, user));
        SM_HIP(hipStreamWaitEvent(lane, plan->ev_inputs, 0));
        if (plan->unfenced) SM_HIP(hipStreamWaitEvent(plan->lane[b ^ 1], plan->ev_inputs, 0));
        plan->unfenced = 0;
    }
    if (inputs_ready) SM_HIP(hipStreamWaitEvent(lane, inputs_ready, 0));
    if (plan->ev_free_set[(q - 3) & 3]) SM_HIP(hipStreamWaitEvent(lane, plan->ev_free[(q - 3) & 3], 0));
    if (shared && plan->ev_free_set[(q - 1) & 3]) SM_HIP(hipStreamWaitEvent(lane, plan->ev_free[(q - 1) & 3], 0));
    plan->seq = q;
    plan->cur = b;
    plan->d_ext = plan->d_ext_buf[b];
    for (int i = 0; i < 2; i++) { plan->out_lo[i] = lo[i]; plan->out_hi[i] = hi[i]; }
    SM_TRY(sm_find_edges(plan, d_gray_left, d_gray_right, threshold, pairs, nullptr, nullptr, (void *)lane));
    SM_TRY(sm_match_wta_typed(plan, pairs, d_web, web_type, d_best, (void *)lane));   // records ev_free[q & 3] on the lane
    SM_HIP(hipStreamWaitEvent(user, plan->ev_free[q & 3], 0));
    return SM_OK;
}

extern "C" int sm_debug_planes(sm_plan *plan, int pair, int shift, uint8_t *d_match,
                               int32_t *d_score_all, int32_t *d_scores, void *stream)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_debug_planes: plan is NULL");
    if (pair < 0 || pair >= plan->pairs_loaded)
        return sm_fail(SM_ERR_ARG, "sm_debug_planes: pair %d not loaded (%d loaded)", pair,
                       plan->pairs_loaded);
    if (shift < 0 || shift >= plan->num_shifts)
        return sm_fail(SM_ERR_ARG, "sm_debug_planes: shift %d outside 0..%d", shift,
                       plan->num_shifts - 1);
    SM_TRY(use_device(plan->device));
    const MatchGeom &g = plan->g;
    const dim3 grid((g.w + 255) / 256, g.h), block(256);
    hipLaunchKernelGGL(k_debug_planes, grid, block, 0, (hipStream_t)stream, plan->d_ext, pair, shift,
                       d_match, d_score_all, d_scores, g, plan->border == SM_GHOST ? 1 : 0);
    SM_LAUNCH_CHECK("k_debug_planes");
    return SM_OK;
}

extern "C" int sm_debug_edge_table(int device, double threshold, uint8_t *d_table, void *stream)
{
    if (!d_table) return sm_fail(SM_ERR_ARG, "sm_debug_edge_table: d_table is NULL");
    SM_TRY(use_device(device));
    hipLaunchKernelGGL(k_edge_table, dim3(3, 766), dim3(256), 0, (hipStream_t)stream, threshold,
                       d_table);
    SM_LAUNCH_CHECK("k_edge_table");
    return SM_OK;
}

extern "C" int sm_fill_web_holes(sm_plan *plan, int32_t *d_web, int32_t *d_tmp, int times,
                                 int pairs, int *result_in_tmp, void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_fill_web_holes"));
    if (!d_web || !d_tmp || !result_in_tmp)
        return sm_fail(SM_ERR_ARG, "sm_fill_web_holes: NULL argument");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)plan->width * plan->height;
    *result_in_tmp = 0;
    if (times <= 0) return SM_OK;

    // The sweeps only ever change pixels that are 0.  The web the hot path
    // produces is >= 1 everywhere (a winning shift is recorded as shift+1), so
    // in the pipeline this stage is the identity; one
    // pass over the image decides that, as the reference's array_min_gpu
    // round trip does for the contour stage.
    SM_HIP(hipMemsetAsync(&plan->d_flags[1], 0, sizeof(i32), st));
    hipLaunchKernelGGL(k_count_zeros, dim3(1024), dim3(256), 0, st, d_web, n * pairs,
                       &plan->d_flags[1]);
    SM_LAUNCH_CHECK("k_count_zeros");
    i32 f[4];
    SM_TRY(read_flags(plan, st, 0, f));
    if (!f[1]) return SM_OK;

    return run_sweeps(plan, d_web, d_tmp, times, pairs, result_in_tmp, st);
}

// one workgroup of 4 waves per 64 K pixels, at most 1024 of them per pair: enough loads in
// flight to stream from HBM, few enough atomics
static dim3 minmax_grid(long long n, int pairs)
{
    long long blocks = (n + 65535) / 65536;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    return dim3((unsigned)blocks, pairs);
}

extern "C" int sm_min_max(sm_plan *plan, const int32_t *d_image, int pairs, int32_t *d_minmax,
                          void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_min_max"));
    if (!d_image || !d_minmax) return sm_fail(SM_ERR_ARG, "sm_min_max: NULL argument");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)plan->width * plan->height;
    hipLaunchKernelGGL(k_step3_init, dim3((pairs + 63) / 64), dim3(64), 0, st, d_minmax, pairs, (i32 *)nullptr);
    hipLaunchKernelGGL(k_minmax_zero, minmax_grid(n, pairs), dim3(256), 0, st, d_image, n, d_minmax,
                       (i32 *)nullptr);
    SM_LAUNCH_CHECK("k_minmax_zero");
    return SM_OK;
}

extern "C" int sm_draw_contour_map(sm_plan *plan, const int32_t *d_web, const int32_t *d_minmax,
                                   int num_lines, int pairs, uint8_t *d_out, void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_draw_contour_map"));
    if (!d_web || !d_minmax || !d_out)
        return sm_fail(SM_ERR_ARG, "sm_draw_contour_map: NULL argument");
    SM_TRY(use_device(plan->device));
    const long long n = (long long)plan->width * plan->height;
    hipLaunchKernelGGL(k_contour, dim3((unsigned)((n + 255) / 256), pairs), dim3(256), 0,
                       (hipStream_t)stream, d_web, d_minmax, num_lines, n, d_out, plan->d_flags);
    SM_LAUNCH_CHECK("k_contour");
    return SM_OK;
}

static int run_sweeps(sm_plan *plan, i32 *d_web, i32 *d_tmp, int times, int pairs, int *result_in_tmp,
                      hipStream_t st)
{
    // tmp <- web, then `times` sweeps with the two buffers trading places
    // (src/stereo.cu:247-256,:328)
    const long long n = (long long)plan->width * plan->height;
    SM_HIP(hipMemcpyAsync(d_tmp, d_web, sizeof(i32) * n * pairs, hipMemcpyDeviceToDevice, st));
    i32 *cur = d_web, *oth = d_tmp;
    const dim3 grid((unsigned)((n + 255) / 256), pairs), block(256);
    for (int i = 0; i < times; i++) {
        hipLaunchKernelGGL(k_fill_holes_step, grid, block, 0, st, cur, oth, plan->width, n);
        i32 *t = cur; cur = oth; oth = t;
    }
    SM_LAUNCH_CHECK("k_fill_holes_step");
    *result_in_tmp = cur == d_tmp;
    return SM_OK;
}

extern "C" int sm_step3(sm_plan *plan, int32_t *d_web, int32_t *d_tmp, int times, int num_lines,
                        int pairs, int32_t *d_minmax, uint8_t *d_out, int *result_in_tmp, void *stream)
{
    SM_TRY(check_plan_pairs(plan, pairs, "sm_step3"));
    if (!d_web || !d_tmp || !d_minmax || !d_out || !result_in_tmp)
        return sm_fail(SM_ERR_ARG, "sm_step3: NULL argument");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)plan->width * plan->height;
    *result_in_tmp = 0;
    const dim3 mm_grid = minmax_grid(n, pairs);
    const dim3 px_grid((unsigned)((n + 255) / 256), pairs);

    // Speculate that the map has no zero pixel (a web from the hot path never has: a
    // winning shift is stored as shift + 1): then hole filling is the identity, the
    // min/max pass over the unfilled map is the one the contour stage needs, and ONE
    // pass also proves the speculation.  Everything is queued before the only sync.
    hipLaunchKernelGGL(k_step3_init, dim3((pairs + 63) / 64), dim3(64), 0, st, d_minmax, pairs, &plan->d_flags[1]);
    hipLaunchKernelGGL(k_minmax_zero, mm_grid, dim3(256), 0, st, d_web, n, d_minmax, &plan->d_flags[1]);
    hipLaunchKernelGGL(k_contour, px_grid, dim3(256), 0, st, d_web, d_minmax, num_lines, n, d_out,
                       plan->d_flags);
    SM_LAUNCH_CHECK("k_contour");
    i32 f[4];
    SM_TRY(read_flags(plan, st, 1, f));
    if (f[1] && times > 0) {
        // there ARE holes: do it the long way (sweeps, then min/max and contour again)
        SM_TRY(run_sweeps(plan, d_web, d_tmp, times, pairs, result_in_tmp, st));
        const i32 *filled = *result_in_tmp ? d_tmp : d_web;
        hipLaunchKernelGGL(k_step3_init, dim3((pairs + 63) / 64), dim3(64), 0, st, d_minmax, pairs, (i32 *)nullptr);
        hipLaunchKernelGGL(k_minmax_zero, mm_grid, dim3(256), 0, st, filled, n, d_minmax, (i32 *)nullptr);
        hipLaunchKernelGGL(k_contour, px_grid, dim3(256), 0, st, filled, d_minmax, num_lines, n, d_out,
                           plan->d_flags);
        SM_LAUNCH_CHECK("k_contour");
        SM_TRY(read_flags(plan, st, 1, f));
    }
    if (f[0])
        return sm_fail(SM_ERR_ZERO_DIV, "contour interval is zero ((max-min)/lines == 0): the "
                       "reference divides by it");
    return SM_OK;
}

extern "C" int sm_plan_status(sm_plan *plan, void *stream)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_plan_status: plan is NULL");
    SM_TRY(use_device(plan->device));
    hipStream_t st = (hipStream_t)stream;
    i32 f[4];
    SM_TRY(read_flags(plan, st, 1, f));
    if (f[0])
        return sm_fail(SM_ERR_ZERO_DIV, "contour interval is zero ((max-min)/lines == 0): the "
                       "reference divides by it");
    return SM_OK;
}
