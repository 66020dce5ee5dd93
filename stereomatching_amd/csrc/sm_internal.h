// sm_internal.h -- shared between the translation units of libstereo_hip.so.
// Not part of the public boundary (that is include/stereo_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "stereo_hip.h"

typedef uint32_t u32;
typedef uint8_t u8;
typedef int32_t i32;

// ---------------------------------------------------------------------------
// Packed edge image ("ext image"), the hot path's input format in HBM.
//
// One bit per pixel, 32 pixels per u32 word, bit b of word k of ext row e is
// the edge value at image coordinate
//        x = 32*k + b - pad_l,        y = e - half.
// The image is stored with its border already applied, so the hot kernel is
// border-mode agnostic for everything except the window's validity mask:
//   toroidal: ext(x, y) = E(x mod W, y mod H)       (idx(), src/util.h:42-47)
//   ghost:    ext(x, y) = E(x, y) inside the image, 0 outside
//                                              (src/stereo-ghost.c:286-287)
// Rows cover y in [-half, rows_needed) and x in [-pad_l, cols_needed) where
// the needed extents include the D-shift, the window halo and tile round-up.
// Workspace layout: [pair][side: 0 = left, 1 = right][ext_rows][ext_words].
// ---------------------------------------------------------------------------

enum { SM_KERNEL_A = 0, SM_KERNEL_B = 1, SM_KERNEL_C = 2, SM_KERNEL_GENERIC = 3, SM_KERNEL_BS = 4 };

#define SM_DSET 16   // shifts per lane in the tiled kernels
#define SM_P 8       // pixels per lane
#define SM_PADT 32   // left pad of a tile's LDS rows, in pixels
#define SM_KEY_DBITS 10  // low bits of the winner key hold the shift

struct MatchGeom {
    int w, h;            // image size
    int D;               // number of shifts
    int n, half;         // window side (odd) and its half
    int ext_words;       // u32 words per ext row
    int ext_rows;        // ext rows per image
    long long ext_image_words;  // ext_words * ext_rows
    int pad_l;           // pixels of left pad in the ext image (multiple of 32)
    // tiled kernels only
    int tile_h;          // output rows per workgroup
    int tw;              // output columns per workgroup (= 8 * runs)
    int runs;            // pixel runs (of 8) per workgroup
    int ds;              // shifts per lane (16; 8 or 16 in the bit-sliced kernel)
    int nl, log2nl;      // lanes that split the shift range of one run
    int threads;         // runs * nl
    int plw, prw;        // words per staged LDS row, left / right
    int nsr;             // staged rows = tile_h + n - 1
    int tiles_x, tiles_y;
    int vec_ok;          // rows are 16-byte aligned -> int4 stores
    int lds_bytes;
    int cap2;            // bit-sliced kernel: launch the two-waves-per-SIMD variant
    int duo;             // bit-sliced kernel: two-wave workgroups of 2 * tile_h rows (shared warm-up)
    unsigned prio_pattern;   // bit-sliced kernel: the time-sliced priority schedule (sm_match_bs_kernel.h)
    int prio_unit;           // ... log2 of the schedule's unit in shader-clock cycles (14: 16384 cycles, ~8 us)
    int prio_shift;          // ... the HW_ID bit that tells a SIMD's two waves apart: 0 wave slot, 16 workgroup slot (TG_ID)
    int prio_on_change;      // ... s_setprio only when the wanted priority changes (else once per row)
    int xmerge;          // bit-sliced kernel: the shift lanes of a word are merged through LDS every 4 rows (nl >= 4)
    int xm_off;          // ... word offset in LDS where the exchange slots of a two-wave workgroup meet and the
                         //     merge buffers lie (wave 0's from here up, wave 1's from here down; a lone wave's from here up)
    int xm_words;        // ... words of one wave's merge buffer
    int web_bytes;       // bytes per element of the web map of THIS launch: 4 (int32), 2, 1
    // ext words per row that can reach a valid output pixel (left image: columns up to W - 1 + half;
    // right: + D - 1 more); the edge kernels compute no others (the tile round-up stays zero)
    int edge_words_l, edge_words_r;
};

struct sm_plan {
    int device;
    sm_plan_options opt;     // sm_plan_create_ex: explicit variant choices (all 0 = the plan's own)
    int width, height, num_shifts, square_width, border, max_pairs;
    int kernel;          // SM_KERNEL_*
    MatchGeom g;
    u32 *d_ext;          // packed edge images: the buffer in use (one of d_ext_buf)
    u32 *d_ext_buf[2];   // double buffer, so that sm_run can pipeline consecutive calls
    size_t ext_bytes;    // of one buffer
    int cur;             // index of d_ext in d_ext_buf
    int pipelined;       // sm_plan_set_pipelined: 0 off, 1 on, 2 on + inputs ordered behind `stream`
    hipStream_t lane[2];         // internal: pipelined sm_run i runs (edges, match) on lane i & 1, into buffer i & 1
    hipEvent_t ev_free[4];       // pipelined call number q has finished: ev_free[q & 3]
    hipEvent_t ev_inputs;        // the caller's stream up to this sm_run (pipelined == 2, or after a sequential phase)
    hipEvent_t ev_fork[2];       // pipelined sm_run INSIDE A STREAM CAPTURE: where lane b leaves the capturing stream
    unsigned long long cap_id;   // ... the capture these belong to (hipStreamGetCaptureInfo)
    int cap_live;                // ... the last pipelined sm_run was captured
    int ev_free_set[4];
    unsigned seq;                // number of the current / last pipelined sm_run
    int unfenced;                // match launches went out without a release event
    uintptr_t out_lo[2], out_hi[2];   // [web, best]: what the last pipelined call writes
    // optional timing of the match launches (sm_plan_time_kernels)
    int timing_cap, timing_n, timing_every, timing_seen;
    hipEvent_t *t_begin, *t_end;
    i32 *d_web_tmp;      // int32 map for narrow results of kernels without a narrow store path
                         // (allocated with the plan for the kernels that need it; part of the workspace)
    i32 *d_flags;        // [0] = zero-interval flag, [1] = has-zero scratch,
                         // [2] = edge table is not of threshold form
    i32 *h_flags;        // pinned host copy of d_flags (k_publish_flags)
    u32 *d_edge_tab;     // 766 x {lo | hi << 16}: edge iff sb <= lo || sb >= hi
    double tab_threshold;  // threshold d_edge_tab was built for
    int tab_valid;
    int tab_ok;          // tables verified to be of threshold form
    int pairs_loaded;    // batch size of the edges currently in d_ext
    char describe[512];
};

// XCD-aware tile order (device side).  Workgroups are dealt round-robin to the 8
// XCDs by their linear id, so ids b and b+8 share an L2.  Mapping linear id ->
// tile so that every XCD owns ONE contiguous run of tiles (row-major: a band of
// image rows) lets the halo rows neighbouring tiles share, and the rows of the
// packed image themselves, be fetched into one L2 instead of eight.  Bijective for
// any tile count (MI355X guide, T1); placement is a speed matter only.
#ifdef __HIPCC__
// linear workgroup id -> position in an XCD-contiguous order over n items
__device__ __forceinline__ int sm_xcd_order(int lin, int n)
{
    const int xcd = lin & 7, j = lin >> 3;
    const int q = n >> 3, r = n & 7;                  // r XCDs get q + 1 items
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
__device__ __forceinline__ void sm_xcd_tile(int tiles_x, int tiles_y, int &tx, int &ty)
{
    const int t = sm_xcd_order(blockIdx.y * gridDim.x + blockIdx.x, tiles_x * tiles_y);
    ty = t / tiles_x;
    tx = t - ty * tiles_x;
}
#endif

// error plumbing (sm_api.hip)
int sm_fail(int code, const char *fmt, ...);
#define SM_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess)                                                 \
            return sm_fail(e_ == hipErrorOutOfMemory ? SM_ERR_NOMEM : SM_ERR_HIP, \
                           "%s failed: %s (%s:%d)", #call,                    \
                           hipGetErrorString(e_), __FILE__, __LINE__);        \
    } while (0)
#define SM_LAUNCH_CHECK(name)                                                 \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess)                                                 \
            return sm_fail(SM_ERR_HIP, "launch of %s failed: %s", name,       \
                           hipGetErrorString(e_));                            \
    } while (0)

// sm_match_bs.hip (bit-sliced kernel; nullptr if not built for this window)
const void *sm_bs_kernel_ptr(int n, int ds, bool fulld, bool ghost, bool cap2, bool duo = false);
int sm_bs_default_ds(int n);
unsigned sm_bs_default_pattern(bool duo);
// what ONE launch adds to the plan's geometry: passed by value, the plan is not modified
struct MatchLaunch {
    MatchGeom g;                        // plan->g with vec_ok / web_bytes of this launch
    hipEvent_t ev_begin, ev_end;        // non-null: attach these events to the dispatch itself
};
int sm_bs_launch(const sm_plan *plan, const MatchLaunch &l, int pairs, i32 *d_web, i32 *d_best, hipStream_t st);
int sm_bs_prepare(sm_plan *plan);        // set-up launch: code object loaded before the first real one

// sm_match.hip
int sm_match_configure(sm_plan *plan);   // fills plan->kernel / plan->g
int sm_match_launch(const sm_plan *plan, const MatchLaunch &l, int pairs, i32 *d_web, i32 *d_best, hipStream_t st);
