// sm_cost.hip -- SAD / SSD cost mode of the hot path.
//
// PARITY UNPINNED: the reference has no SAD/SSD implementation (SURVEY.md
// section 0); BASELINE.json merely words the hot path that way.  This mode is
// the build's own definition on the same skeleton (per-shift cost -> n x n box
// sum -> winner-take-all), checked only against the build's own CPU definition
// of the mode (the checker's cost_hot_path):
//   c_d(x,y) = |L(x,y) - R(x+d,y)| (SAD) or its square (SSD) on the uint8 gray
//   images; R wraps (toroidal) or reads 0 past the right border (ghost);
//   A_d = window sum (wrapping, or taps outside the image counting 0);
//   best = min_d A_d, web = 1 + the FIRST d reaching it.
//
// Byte arithmetic, so no bit-slicing here: a lane owns 4 consecutive pixels x 8
// shifts and marches down a tile with 32 window sums in VGPRs.  Per row and
// shift the right window (n + 3 bytes) is cut out of the staged row with
// v_alignbyte_b32, and each pixel's n-byte window inside it is selected by byte
// masks: SAD is v_sad_u8 with accumulate on the masked dwords; SSD is
// LL + RR - 2*LR from v_dot4_u32_u8.  The old row leaves the sums the same way.
// The shift range of a pixel group is split over nl = D/8 adjacent lanes whose
// (sum, shift) winners are merged with DPP.  HBM side: coalesced row loads of
// the two gray images into LDS (border rule applied while staging).

#include "sm_internal.h"
#include "sm_cost.h"
#include <mutex>

#define SMC_DS 8        // shifts per lane
#define SMC_PX 4        // pixels per lane

template <int CTRL>
__device__ __forceinline__ u32 dppc(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// partner value for merge step K of the nl-lane group
template <int K>
__device__ __forceinline__ u32 partner(u32 v)
{
    if (K == 0) return dppc<0xB1>(v);
    if (K == 1) return dppc<0x4E>(v);
    if (K == 2) return dppc<0x141>(v);
    if (K == 3) return dppc<0x140>(v);
    return (u32)__shfl_xor((int)v, 1 << K);
}

struct CostGeom {
    int w, h, D, n, half;
    int tile_h, tw, groups, nl, log2nl;
    int lrow, rrow;          // bytes per staged row (multiples of 4)
    int nsr, tiles_x, tiles_y;
    int pad;                 // bytes left of the tile in a staged row (multiple of 4, >= half)
    int vec_ok;
    int xlim;                // only pixel groups left of this column are computed (the whole image: w)
};

template <int NWD, bool SSD, bool GHOST>
__global__ __launch_bounds__(256) void k_cost_wta(const u8 *__restrict__ left,
                                                  const u8 *__restrict__ right,
                                                  i32 *__restrict__ web, i32 *__restrict__ best,
                                                  const CostGeom g)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds_b[];
    const int tid = threadIdx.x;
    const int pair = blockIdx.z;
    const int tx0 = blockIdx.x * g.tw, ty0 = blockIdx.y * g.tile_h;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *L = left + img, *R = right + img;
    u8 *sL = lds_b;
    u8 *sR = sL + (size_t)g.nsr * g.lrow;
    const int n = g.n, half = g.half;

    // ---- stage rows ty0-half .. with the border rule applied.  Eight rows at a time: the eight byte
    // loads of a lane are all issued before the first is stored (one row per trip paid one memory round
    // trip per row and column block: 72 trips in the ghost strip's 64-thread workgroups, 60 of their 70 us)
    constexpr int SR = 8;
    for (int row0 = 0; row0 < g.nsr; row0 += SR) {
        for (int b = tid; b < g.lrow + g.rrow; b += (int)blockDim.x) {
            const bool is_r = b >= g.lrow;
            const int bb = is_r ? b - g.lrow : b;
            const int x = tx0 - g.pad + bb;
            // (every load is unconditional, from an address clamped into the image: a load under a
            // condition is waited for at the end of its branch)
            const int xs = GHOST ? min(max(x, 0), g.w - 1) : ((x % g.w) + g.w) % g.w;
            const bool vx = !GHOST || (x >= 0 && x < g.w);
            const u8 *src = is_r ? R : L;
            u8 v[SR];
#pragma unroll
            for (int u = 0; u < SR; u++) {
                const int y = ty0 - half + row0 + u;
                const bool vy = y >= 0 && y < g.h;
                const int ys = GHOST ? min(max(y, 0), g.h - 1) : ((y % g.h) + g.h) % g.h;
                const u8 raw = src[(size_t)ys * g.w + xs];
                v[u] = (vx && (vy || !GHOST)) ? raw : (u8)0;
            }
#pragma unroll
            for (int u = 0; u < SR; u++)
                if (row0 + u < g.nsr)
                    (is_r ? sR + (size_t)(row0 + u) * g.rrow : sL + (size_t)(row0 + u) * g.lrow)[bb] = v[u];
        }
    }
    __syncthreads();

    // ---- lane role
    const int s = tid & (g.nl - 1);
    const int grp = tid >> g.log2nl;
    const int x0l = grp * SMC_PX, x0 = tx0 + x0l;
    const int d0 = s * SMC_DS;
    if (x0 >= g.xlim) return;                      // (all lanes of a pixel group leave together)
    const int bL = g.pad + x0l - half;            // byte offset of the window in a staged row
    const int wL = bL >> 2, shL = bL & 3;
    const int bR = bL + d0;
    const int wR = bR >> 2, shR = bR & 3;         // shR == shL (d0 is a multiple of 8)

    // byte masks: pixel j's window is bytes j .. j+n-1 of the NWD-dword window;
    // ghost: columns outside the image are masked out as well
    u32 M[SMC_PX][NWD];
#pragma unroll
    for (int j = 0; j < SMC_PX; j++)
#pragma unroll
        for (int k = 0; k < NWD; k++) {
            u32 m = 0;
            for (int b = 0; b < 4; b++) {
                const int byte = 4 * k + b;
                bool in = byte >= j && byte < j + n;
                if (GHOST) { const int x = x0 - half + byte; in = in && x >= 0 && x < g.w; }
                if (in) m |= 0xffu << (8 * b);
            }
            M[j][k] = m;
        }

    u32 A[SMC_DS][SMC_PX];
#pragma unroll
    for (int dd = 0; dd < SMC_DS; dd++)
#pragma unroll
        for (int j = 0; j < SMC_PX; j++) A[dd][j] = 0;

    auto slide = [&](int srow, bool add) {
        const u32 *rl = reinterpret_cast<const u32 *>(sL + (size_t)srow * g.lrow) + wL;
        const u32 *rr = reinterpret_cast<const u32 *>(sR + (size_t)srow * g.rrow) + wR;
        u32 lm[SMC_PX][NWD], ll[SMC_PX];
#pragma unroll
        for (int k = 0; k < NWD; k++) {
            const u32 lw = __builtin_amdgcn_alignbyte(rl[k + 1], rl[k], shL);
#pragma unroll
            for (int j = 0; j < SMC_PX; j++) lm[j][k] = lw & M[j][k];
        }
        if (SSD) {
#pragma unroll
            for (int j = 0; j < SMC_PX; j++) {
                u32 acc = 0;
#pragma unroll
                for (int k = 0; k < NWD; k++) acc = __builtin_amdgcn_udot4(lm[j][k], lm[j][k], acc, false);
                ll[j] = acc;
            }
        }
        u32 raw[NWD + 3];
#pragma unroll
        for (int k = 0; k < NWD + 3; k++) raw[k] = rr[k];
#pragma unroll
        for (int dd = 0; dd < SMC_DS; dd++) {
            // right window at byte offset shR + dd of raw[]
            u32 rw[NWD];
#pragma unroll
            for (int k = 0; k < NWD; k++) {
                const int q = (shR + dd) >> 2;        // uniform
                rw[k] = __builtin_amdgcn_alignbyte(raw[k + q + 1], raw[k + q], (shR + dd) & 3);
            }
#pragma unroll
            for (int j = 0; j < SMC_PX; j++) {
                u32 v;
                if (SSD) {
                    u32 rrs = 0, lr = 0;
#pragma unroll
                    for (int k = 0; k < NWD; k++) {
                        const u32 rm = rw[k] & M[j][k];
                        rrs = __builtin_amdgcn_udot4(rm, rm, rrs, false);
                        lr = __builtin_amdgcn_udot4(lm[j][k], rm, lr, false);
                    }
                    v = ll[j] + rrs - 2u * lr;
                } else {
                    v = 0;
#pragma unroll
                    for (int k = 0; k < NWD; k++) v = __builtin_amdgcn_sad_u8(lm[j][k], rw[k] & M[j][k], v);
                }
                if (add) A[dd][j] += v; else A[dd][j] -= v;
            }
        }
    };

    const int rows_out = min(g.tile_h, g.h - ty0);
    const int steps = rows_out + n - 1;
    const int dlim = g.D - d0;                        // shifts of this lane below D
    for (int e = 0; e < steps; e++) {
        const int y_new = ty0 - half + e;
        if (!GHOST || (y_new >= 0 && y_new < g.h)) slide(e, true);
        if (e >= n) {
            const int y_old = y_new - n;
            if (!GHOST || (y_old >= 0 && y_old < g.h)) slide(e - n, false);
        }
        if (e < n - 1) continue;
        const int y = ty0 + e - (n - 1);

        // first-wins arg-min over this lane's shifts, then over the nl lanes
        u32 ba[SMC_PX], bd[SMC_PX];
#pragma unroll
        for (int j = 0; j < SMC_PX; j++) { ba[j] = 0xffffffffu; bd[j] = 0xffffffffu; }
#pragma unroll
        for (int dd = 0; dd < SMC_DS; dd++) {
            if (dd < dlim) {
#pragma unroll
                for (int j = 0; j < SMC_PX; j++)
                    if (A[dd][j] < ba[j]) { ba[j] = A[dd][j]; bd[j] = (u32)(d0 + dd); }
            }
        }
#define SMC_MERGE(K)                                                            \
        if (g.nl > (1 << K)) {                                                  \
            _Pragma("unroll") for (int j = 0; j < SMC_PX; j++) {                \
                const u32 pa = partner<K>(ba[j]), pd = partner<K>(bd[j]);       \
                const bool take = pa < ba[j] || (pa == ba[j] && pd < bd[j]);    \
                ba[j] = take ? pa : ba[j];                                      \
                bd[j] = take ? pd : bd[j];                                      \
            }                                                                   \
        }
        SMC_MERGE(0) SMC_MERGE(1) SMC_MERGE(2) SMC_MERGE(3) SMC_MERGE(4) SMC_MERGE(5)
#undef SMC_MERGE

        if (s == 0 && x0 < g.w) {
            const size_t o = ((size_t)pair * g.h + y) * g.w + x0;
            if (g.vec_ok && x0 + SMC_PX <= g.w) {
                *reinterpret_cast<int4 *>(web + o) =
                    make_int4((i32)bd[0] + 1, (i32)bd[1] + 1, (i32)bd[2] + 1, (i32)bd[3] + 1);
                if (best)
                    *reinterpret_cast<int4 *>(best + o) =
                        make_int4((i32)ba[0], (i32)ba[1], (i32)ba[2], (i32)ba[3]);
            } else {
#pragma unroll
                for (int j = 0; j < SMC_PX; j++)
                    if (x0 + j < g.w) {
                        web[o + j] = (i32)bd[j] + 1;
                        if (best) best[o + j] = (i32)ba[j];
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------

template <int NWD>
static const void *cost_ptr(bool ssd, bool ghost)
{
    return ssd ? (ghost ? (const void *)k_cost_wta<NWD, true, true> : (const void *)k_cost_wta<NWD, true, false>)
               : (ghost ? (const void *)k_cost_wta<NWD, false, true> : (const void *)k_cost_wta<NWD, false, false>);
}

// the general (masked) kernel on the whole image, or -- strip_cols > 0 -- only on the pixel
// columns [0, strip_cols) (the ghost-border columns the quad-SAD kernel cannot do)
static int launch_general(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right, int cost,
                          int pairs, int32_t *d_web, int32_t *d_best, int strip_cols, hipStream_t stream)
{
    CostGeom g;
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts;
    g.half = plan->square_width / 2; g.n = 2 * g.half + 1;
    if (g.n > 25 || g.D > 512)
        return sm_fail(SM_ERR_ARG, "sm_cost_wta: built for windows up to 25x25 and at most 512 shifts "
                       "(got %dx%d, %d)", g.n, g.n, g.D);
    g.nl = 1; g.log2nl = 0;
    while (g.nl * SMC_DS < g.D) { g.nl <<= 1; g.log2nl++; }
    // whole image: 256 threads = 256 / nl pixel groups per workgroup; the ghost strip: only the
    // groups that hold its few columns (a narrow tile: little to stage), rounded up to whole waves
    int threads = 256;
    g.groups = 256 / g.nl;
    if (strip_cols > 0) {
        const int need = (strip_cols + SMC_PX - 1) / SMC_PX;
        if (need < g.groups) g.groups = need;
        threads = 64 * ((g.groups * g.nl + 63) / 64);
        g.groups = threads / g.nl;
    }
    g.tw = g.groups * SMC_PX;
    g.pad = 4 * ((g.half + 3) / 4);
    int nwd = (g.n + 3 + 3) / 4;       // dwords holding the n + 3 window bytes of a lane
    if (nwd < 2) nwd = 2;              // smallest instantiation
    // bytes read by the last group: L: pad + tw - 4 - half + 4*(nwd+1); R additionally nl*8 + 12
    g.lrow = 4 * ((g.pad + g.tw + 4 * (nwd + 1) + 3) / 4);
    g.rrow = 4 * ((g.pad + g.tw + g.nl * SMC_DS + 4 * (nwd + 4) + 3) / 4);
    g.tiles_x = (g.w + g.tw - 1) / g.tw;
    g.xlim = g.w;
    if (strip_cols > 0) {
        g.xlim = strip_cols;
        g.tiles_x = (strip_cols + g.tw - 1) / g.tw;
    }
    int th = 64;
    // (the strip is a few hundred short workgroups whatever the tile height: its duration is one
    // workgroup's latency, so it takes the shortest tiles)
    while (th > (strip_cols > 0 ? 2 : 8) && (long long)g.tiles_x * ((g.h + th - 1) / th) * pairs < (strip_cols > 0 ? 4096 : 1024)) th >>= 1;
    while ((th + g.n - 1) * (g.lrow + g.rrow) > 60 * 1024 && th > 1) th >>= 1;
    th = th < g.h ? th : g.h;
    g.tile_h = th;
    g.tiles_y = (g.h + th - 1) / th;
    g.nsr = th + g.n - 1;
    g.vec_ok = g.w % 4 == 0;
    const bool ssd = cost == SM_COST_SSD, ghost = plan->border == SM_GHOST;
    const void *fn;
    switch (nwd) {
    case 2: fn = cost_ptr<2>(ssd, ghost); break;
    case 3: fn = cost_ptr<3>(ssd, ghost); break;
    case 4: fn = cost_ptr<4>(ssd, ghost); break;
    case 5: fn = cost_ptr<5>(ssd, ghost); break;
    case 6: fn = cost_ptr<6>(ssd, ghost); break;
    default: fn = cost_ptr<7>(ssd, ghost); break;
    }
    void *args[] = {(void *)&d_gray_left, (void *)&d_gray_right, (void *)&d_web, (void *)&d_best, (void *)&g};
    const hipError_t e = hipLaunchKernel(fn, dim3(g.tiles_x, g.tiles_y, pairs), dim3(threads), args,
                                         (size_t)g.nsr * (g.lrow + g.rrow), stream);
    if (e != hipSuccess) return sm_fail(SM_ERR_HIP, "sm_cost_wta: %s", hipGetErrorString(e));
    return SM_OK;
}

extern "C" int sm_cost_wta(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                           int cost, int pairs, int32_t *d_web, int32_t *d_best, void *stream)
{
    if (!plan) return sm_fail(SM_ERR_ARG, "sm_cost_wta: plan is NULL");
    if (pairs < 1 || pairs > plan->max_pairs)
        return sm_fail(SM_ERR_ARG, "sm_cost_wta: pairs %d outside 1..%d", pairs, plan->max_pairs);
    if (!d_gray_left || !d_gray_right || !d_web) return sm_fail(SM_ERR_ARG, "sm_cost_wta: NULL argument");
    if (cost != SM_COST_SAD && cost != SM_COST_SSD)
        return sm_fail(SM_ERR_ARG, "sm_cost_wta: cost %d is neither SM_COST_SAD nor SM_COST_SSD", cost);
    const hipError_t es = hipSetDevice(plan->device);
    if (es != hipSuccess) return sm_fail(SM_ERR_HIP, "sm_cost_wta: %s", hipGetErrorString(es));
    {
        // SAD on the quad-SAD unit (sm_cost_pc.hip: window rows by prefix chains, windows up to 15 x 15; sm_cost_qs.hip:
        // the larger ones, and cost_kernel = 4), SSD on the matrix cores (sm_cost_mfma.hip; cost_kernel = 2: on the
        // byte dot-product unit, sm_cost_ssd.hip),
        // where they are built for this window and shift count; otherwise the general masked kernel
        SadGeom q;
        const void *fn = cost == SM_COST_SAD ? sm_sad_pc_configure(plan, pairs, d_gray_left, d_gray_right, &q)
                                             : sm_ssd_mfma_configure(plan, pairs, d_gray_left, d_gray_right, &q);
        if (!fn && cost == SM_COST_SAD) fn = sm_sad_qs_configure(plan, pairs, d_gray_left, d_gray_right, &q);
        if (!fn && cost == SM_COST_SSD) fn = sm_ssd_dot_configure(plan, pairs, d_gray_left, d_gray_right, &q);
        if (fn) {
            void *args[] = {(void *)&d_gray_left, (void *)&d_gray_right, (void *)&d_web, (void *)&d_best, (void *)&q};
            if (q.lds_bytes > 64 * 1024) {
                // (four-wave workgroups of k_sad_pc: up to 80 of the CU's 160 KB; the limit is raised once per kernel)
                static std::mutex guard;          // (plans on different host threads may launch the same kernel)
                static const void *raised[8];
                static int n_raised = 0;
                std::lock_guard<std::mutex> lock(guard);
                bool seen = false;
                for (int i = 0; i < n_raised; i++) seen = seen || raised[i] == fn;
                if (!seen) {
                    SM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                    if (n_raised < 8) raised[n_raised++] = fn;
                }
            }
            const hipError_t e = hipLaunchKernel(fn, dim3(q.tiles_x, q.tiles_y, pairs), dim3(64 * q.waves), args,
                                                 (size_t)q.lds_bytes, (hipStream_t)stream);
            if (e != hipSuccess) return sm_fail(SM_ERR_HIP, "sm_cost_wta: %s", hipGetErrorString(e));
            // Ghost border: the columns whose windows reach left of the image (x < half), by the masked kernel,
            // BEHIND the main launch on the same stream.  (Round 4 ran this strip -- a few hundred short
            // workgroups, 50-60 us at 4K -- BESIDE the main launch on a stream of its own, the two writing
            // disjoint columns: the main launch then takes 130 us longer at C5 SAD, 881 vs 754 us, and 23 us at
            // C5 SSD -- the strip's 256-thread workgroups take their CUs' LDS and wave slots first and the main
            // launch's one-wave workgroups no longer spread evenly; enqueued behind the main launch instead of in front
            // of it, the same: 876 vs 754 us.  profiles/r04/ab_cost_strip_beside_rejected.txt)
            if (q.ghost && plan->square_width / 2 > 0) {
                // sm_cost_strip.hip: a kernel built for these columns (~10 us at 4K where the general one needs 50-60)
                const int rc = sm_cost_strip_launch(plan, d_gray_left, d_gray_right, cost, pairs, d_web, d_best,
                                                    (hipStream_t)stream);
                if (rc >= 0) return rc;
                return launch_general(plan, d_gray_left, d_gray_right, cost, pairs, d_web, d_best,
                                      plan->square_width / 2, (hipStream_t)stream);
            }
            return SM_OK;
        }
    }
    return launch_general(plan, d_gray_left, d_gray_right, cost, pairs, d_web, d_best, 0, (hipStream_t)stream);
}
