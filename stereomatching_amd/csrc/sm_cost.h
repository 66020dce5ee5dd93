// sm_cost.h -- shared between the translation units of the SAD / SSD cost mode
// (sm_cost.hip: the general masked kernel and the C entry; sm_cost_qs.hip: SAD on the quad-SAD unit).
#pragma once
#include "sm_internal.h"

struct SadGeom {
    int w, h, D;
    int ghost;
    int tile_h, tw;          // output rows / columns per (one-wave) workgroup
    int nl, log2nl;          // lanes that split the shift range of one pixel group
    int nql, px;             // shift quads and pixels per lane (the kernel's template arguments)
    int tiles_x, tiles_y;
    int padl;                // bytes left of the tile in a staged row (multiple of 4, >= half + 3)
    int lrow, rrow;          // bytes per staged row, left / right (multiples of 8)
    int nsr;                 // staged rows = tile_h + n - 1
    int q_tail;              // first quad of a lane that may hold shifts >= D
    int q_last;              // last quad in which some lane has a shift < D
    int fast_stage;          // image rows are dword-aligned and w % 4 == 0
    int rr_stride;           // k_ssd_dot: dwords between the four residue classes of its RR table
    int tbl_pad;             // k_ssd_mfma: dwords between the staged rows and its (16-byte aligned) RR table
    int lds_bytes;
    int waves;               // k_sad_pc: waves per workgroup (they share the staged rows; the other kernels: 1)
};

#ifdef __HIPCC__
// Stage rows ty0 - half .. ty0 - half + nsr - 1 of one pair's two gray images into LDS, [nsr][lrow]
// bytes of the left image then [nsr][rrow] of the right one, byte 0 of a staged row = image column
// xw - padl, with the border rule applied: wrap-around (toroidal) or zeros outside the image (ghost).
// All `nthreads` threads of the workgroup take part (one wave unless said otherwise).  Shared by the SAD and the SSD kernels.
// `flip` is XORed onto every staged dword (the SSD kernel stages pixel - 128 as signed bytes: 0x80808080).
// `rows`: how many of the g.nsr rows to stage (the others are the caller's: SmcStream below).
__device__ __forceinline__ void smc_stage_rows(u32 *lds, const u8 *__restrict__ L, const u8 *__restrict__ R,
                                               const SadGeom &g, int xw, int ty0, int HALF, int tid, u32 flip = 0,
                                               int nthreads = 64, int rows = 1 << 30)
{
    const int lw = g.lrow >> 2, rw = g.rrow >> 2;
    const int nsr_all = g.nsr;
    if (rows > nsr_all) rows = nsr_all;
    if (g.fast_stage) {
        // Image width a multiple of 4 and dword-aligned rows: a staged dword never straddles a border.
        // A lane's dword columns are the same in every row, so the column arithmetic (the wrap-around
        // or the border test) is done once, and the rows are loaded SR at a time with all loads in
        // flight -- a load, a wait and a store per row and column cost a quarter of the kernel's
        // instructions and left the wave waiting for memory 40 times over.
        constexpr int SC = 4, SR = 4;                   // (lw + rw <= 4 nthreads dwords: checked by the host)
        const u8 *col[SC];
        int dst[SC], dstride[SC];
        bool on[SC];
#pragma unroll
        for (int c = 0; c < SC; c++) {
            const int k = tid + nthreads * c;
            const bool is_r = k >= lw;
            const int kk = is_r ? k - lw : k;
            const int x = xw - g.padl + 4 * kk;
            on[c] = k < lw + rw && (!g.ghost || (x >= 0 && x < g.w));
            const int xs = g.ghost ? x : ((x % g.w) + g.w) % g.w;
            col[c] = (is_r ? R : L) + (on[c] ? xs : 0);
            dst[c] = is_r ? g.nsr * lw + kk : kk;
            dstride[c] = is_r ? rw : lw;
            if (k >= lw + rw) dst[c] = -1;
        }
        for (int row0 = 0; row0 < rows; row0 += SR) {
            u32 v[SR][SC];
#pragma unroll
            for (int r = 0; r < SR; r++) {
                const int y = ty0 - HALF + row0 + r;
                const bool vy = (y >= 0 && y < g.h) || !g.ghost;
                const int ys = g.ghost ? (vy ? y : 0) : ((y % g.h) + g.h) % g.h;
#pragma unroll
                for (int c = 0; c < SC; c++) {
                    v[r][c] = 0;
                    if (on[c] && vy && row0 + r < rows)
                        v[r][c] = *reinterpret_cast<const u32 *>(col[c] + (size_t)ys * g.w);
                }
            }
#pragma unroll
            for (int r = 0; r < SR; r++)
#pragma unroll
                for (int c = 0; c < SC; c++)
                    if (dst[c] >= 0 && row0 + r < rows) lds[dst[c] + (row0 + r) * dstride[c]] = v[r][c] ^ flip;
        }
    } else {
        for (int row = 0; row < rows; row++) {
            const int y = ty0 - HALF + row;
            const bool vy = y >= 0 && y < g.h;
            const int ys = g.ghost ? (vy ? y : 0) : ((y % g.h) + g.h) % g.h;
            for (int k = tid; k < lw + rw; k += nthreads) {
                const bool is_r = k >= lw;
                const int kk = is_r ? k - lw : k;
                const int x = xw - g.padl + 4 * kk;
                const u8 *src = (is_r ? R : L) + (size_t)ys * g.w;
                u32 v = 0;
                for (int b = 0; b < 4; b++) {
                    const int xb = x + b;
                    u32 p = 0;
                    if (g.ghost) { if (vy && xb >= 0 && xb < g.w) p = src[xb]; }
                    else p = src[((xb % g.w) + g.w) % g.w];
                    v |= p << (8 * b);
                }
                (is_r ? lds + g.nsr * lw + row * rw : lds + row * lw)[kk] = v ^ flip;
            }
        }
    }
}

// The rows of a tile fetched WHILE the tile is being worked on (round 5, k_sad_pc): staging all g.nsr rows before the first
// window row is 5 % of a launch (every wave of a one-round launch fetches at the same time and none computes:
// profiles/r05/ab_sad_knockouts.txt).  Only where smc_stage_rows takes its fast path (g.fast_stage): a lane's dword
// columns are the same in every row, so the column arithmetic is done once (here), a row is `SC` loads per lane issued at
// one step (fetch) and as many LDS stores at the next (store) -- the caller's barrier of that step publishes them.
struct SmcStream {
    static constexpr int SC = 4;                        // (lw + rw <= 4 nthreads dwords: checked by the host)
    const u8 *col[SC];
    int dst[SC], dstride[SC];
    bool on[SC];
    u32 v[SC];
    int pending;                                        // row whose dwords are in v (in flight), -1: none

    __device__ __forceinline__ void setup(const u8 *__restrict__ L, const u8 *__restrict__ R, const SadGeom &g, int xw,
                                          int tid, int nthreads)
    {
        const int lw = g.lrow >> 2, rw = g.rrow >> 2;
#pragma unroll
        for (int c = 0; c < SC; c++) {
            const int k = tid + nthreads * c;
            const bool is_r = k >= lw;
            const int kk = is_r ? k - lw : k;
            const int x = xw - g.padl + 4 * kk;
            on[c] = k < lw + rw && (!g.ghost || (x >= 0 && x < g.w));
            const int xs = g.ghost ? x : ((x % g.w) + g.w) % g.w;
            col[c] = (is_r ? R : L) + (on[c] ? xs : 0);
            dst[c] = is_r ? g.nsr * lw + kk : kk;
            dstride[c] = is_r ? rw : lw;
            if (k >= lw + rw) dst[c] = -1;
            v[c] = 0;
        }
        pending = -1;
    }
    // the loads of staged row `row` (image row ty0 - HALF + row, border rule applied)
    __device__ __forceinline__ void fetch(const SadGeom &g, int ty0, int HALF, int row)
    {
        const int y = ty0 - HALF + row;
        const bool vy = (y >= 0 && y < g.h) || !g.ghost;
        const int ys = g.ghost ? (vy ? y : 0) : ((y % g.h) + g.h) % g.h;
#pragma unroll
        for (int c = 0; c < SC; c++) {
            v[c] = 0;
            if (on[c] && vy) v[c] = *reinterpret_cast<const u32 *>(col[c] + (size_t)ys * g.w);
        }
        pending = row;
    }
    __device__ __forceinline__ void store(u32 *lds)
    {
        if (pending < 0) return;
#pragma unroll
        for (int c = 0; c < SC; c++)
            if (dst[c] >= 0) lds[dst[c] + pending * dstride[c]] = v[c];
        pending = -1;
    }
};
#endif

// sm_cost_qs.hip / sm_cost_pc.hip / sm_cost_ssd.hip / sm_cost_mfma.hip: fill *g and return the kernel for this plan, or nullptr if the shape is not built
const void *sm_sad_qs_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *g);
// sm_cost_pc.hip: SAD with the window rows formed by prefix chains along the row (round 5; windows up to 15 x 15)
const void *sm_sad_pc_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *g);
const void *sm_ssd_dot_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *g);
const void *sm_ssd_mfma_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *g);
// sm_cost_strip.hip: the ghost-border columns x < half behind a fast kernel's launch; -1 if not built for this shape
int sm_cost_strip_launch(const sm_plan *plan, const uint8_t *d_left, const uint8_t *d_right, int cost, int pairs,
                         int32_t *d_web, int32_t *d_best, hipStream_t stream);
