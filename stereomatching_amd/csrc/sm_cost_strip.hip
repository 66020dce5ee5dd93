// sm_cost_strip.hip -- the ghost-border strip of the SAD / SSD cost mode: the pixel columns x < half.
//
// PARITY UNPINNED, as the whole cost mode (top of sm_cost.hip).
//
// With the ghost border a window tap outside the image counts 0.  The fast kernels (sm_cost_qs.hip,
// sm_cost_mfma.hip, sm_cost_ssd.hip) stage zero pixels there, which is the same thing wherever BOTH
// images are outside -- above, below and right of the image -- but not left of it: the left pixel is
// outside, the right one at x' + d is not.  Those taps exist for the columns x < half only, a strip of
// half x H pixels (0.13 % of a 4K image at 11 x 11) that the fast kernels leave out.  Until round 4 the
// general masked kernel ran on it: 50-60 us at 4K whatever its tile height -- built for whole images, it
// wastes three of its four pixel groups here and one of its workgroups needs 4 us per row.
//
// This kernel is built for the strip: every window of it starts at column 0, so a pixel's window row is
// the first x + half + 1 bytes of the row -- two or three dwords, the last one masked.  A thread owns one
// shift (two beyond 256) and all `half` pixels, slides their sums down tile_h rows (SAD: v_sad_u8 on the
// masked dwords; SSD: LL + RR - 2 LR from v_dot4_u32_u8), and the 256 shifts of a pixel meet in a wave
// reduction per row plus one pass over the waves' results at the end of the tile.  4K, 11 x 11, 256 shifts:
// ~10 us behind the main launch instead of 61.

#include "sm_internal.h"
#include "sm_cost.h"
#include <type_traits>

struct StripGeom {
    int w, h, D;
    int tile_h, nsr;        // output rows per workgroup, staged rows = tile_h + 2 half
    int rw;                 // dwords per staged right row
};

template <int H, bool SSD, int NS>
__global__ __launch_bounds__(256) void k_cost_strip(const u8 *__restrict__ left, const u8 *__restrict__ right,
                                                    i32 *__restrict__ web, i32 *__restrict__ best,
                                                    const StripGeom g)
{
    constexpr int NWD = (2 * H + 3) / 4;                 // dwords holding the widest window row: 2 H bytes
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int tid = threadIdx.x, pair = blockIdx.z;
    const int y0 = blockIdx.y * g.tile_h;
    const size_t img = (size_t)pair * g.w * g.h;
    const u8 *L = left + img, *R = right + img;
    u32 *sL = lds;                                       // [nsr][NWD]
    u32 *sR = sL + g.nsr * NWD;                          // [nsr][rw]
    u32 *sK = sR + g.nsr * g.rw;                         // [tile_h][H][4 waves]

    // ---- stage: column c of staged row r is pixel (c, y0 - H + r), 0 outside the image (both images)
    const bool aligned = g.w % 4 == 0 && (((uintptr_t)L | (uintptr_t)R) & 3) == 0;
    const int per_row = NWD + g.rw;
    for (int i = tid; i < g.nsr * per_row; i += 256) {
        const int r = i / per_row, k = i - r * per_row;
        const bool is_r = k >= NWD;
        const int kk = is_r ? k - NWD : k;
        const int y = y0 - H + r;
        u32 v = 0;
        if (y >= 0 && y < g.h) {
            const u8 *src = (is_r ? R : L) + (size_t)y * g.w;
            if (aligned && 4 * kk + 3 < g.w) v = *reinterpret_cast<const u32 *>(src + 4 * kk);
            else
                for (int b = 0; b < 4; b++)
                    if (4 * kk + b < g.w) v |= (u32)src[4 * kk + b] << (8 * b);
        }
        (is_r ? sR + r * g.rw : sL + r * NWD)[kk] = v;
    }
    __syncthreads();

    // SAD: A[s][x] the window sum; SSD: A = LR, B = RR of the pixel's (narrower) window, LLs = LL
    u32 A[NS][H], B[NS][H], LLs[H];
#pragma unroll
    for (int x = 0; x < H; x++) {
        LLs[x] = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) { A[s][x] = 0; B[s][x] = 0; }
    }

    auto masked = [](u32 v, int x, int k) -> u32 {      // dword k of pixel x's window row: bytes 0 .. x + H
        const int taps = x + H + 1, fd = taps / 4, rem = taps % 4;
        return k < fd ? v : (k == fd && rem) ? v & ((1u << (8 * rem)) - 1u) : 0u;
    };
    auto ndw = [](int x) { return (x + H + 1 + 3) / 4; };

    // one window row in, or out
    auto feed = [&](int row, auto out_tag) {
        constexpr bool out = decltype(out_tag)::value;
        u32 l[NWD];
#pragma unroll
        for (int k = 0; k < NWD; k++) l[k] = sL[row * NWD + k];
        if (SSD) {
#pragma unroll
            for (int x = 0; x < H; x++) {
                u32 t = 0;
#pragma unroll
                for (int k = 0; k < NWD; k++)
                    if (k < ndw(x)) { const u32 q = masked(l[k], x, k); t = __builtin_amdgcn_udot4(q, q, t, false); }
                LLs[x] += out ? 0u - t : t;
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int d = tid + 256 * s;
            const u32 *rr = sR + row * g.rw + (d >> 2);
            u32 t[NWD + 1], r[NWD];
#pragma unroll
            for (int k = 0; k <= NWD; k++) t[k] = rr[k];
#pragma unroll
            for (int k = 0; k < NWD; k++) r[k] = __builtin_amdgcn_alignbyte(t[k + 1], t[k], (u32)(d & 3));
#pragma unroll
            for (int x = 0; x < H; x++) {
                u32 a = 0, b = 0;
#pragma unroll
                for (int k = 0; k < NWD; k++) {
                    if (k >= ndw(x)) continue;
                    const u32 lm = masked(l[k], x, k), rm = masked(r[k], x, k);
                    if (SSD) {
                        a = __builtin_amdgcn_udot4(lm, r[k], a, false);
                        b = __builtin_amdgcn_udot4(rm, rm, b, false);
                    } else {
                        a = __builtin_amdgcn_sad_u8(lm, rm, a);
                    }
                }
                A[s][x] += out ? 0u - a : a;
                if (SSD) B[s][x] += out ? 0u - b : b;
            }
        }
    };

    const int rows_out = min(g.tile_h, g.h - y0);
    const int wave = tid >> 6;
    // staged row e is image row y0 - H + e: output row t has window rows t .. t + 2 H
    for (int e = 0; e < 2 * H; e++) feed(e, std::false_type{});
    for (int t = 0; t < rows_out; t++) {
        feed(t + 2 * H, std::false_type{});
#pragma unroll
        for (int x = 0; x < H; x++) {
            u32 key = 0xffffffffu;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int d = tid + 256 * s;
                const u32 c = SSD ? LLs[x] + B[s][x] - 2 * A[s][x] : A[s][x];
                if (d < g.D) key = min(key, (c << 9) | (u32)d);     // the lowest cost, among equals the first shift
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) key = min(key, (u32)__shfl_xor((int)key, m));
            if ((tid & 63) == 0) sK[(t * H + x) * 4 + wave] = key;
        }
        feed(t, std::true_type{});
    }
    __syncthreads();
    for (int i = tid; i < rows_out * H; i += 256) {
        const u32 key = min(min(sK[4 * i], sK[4 * i + 1]), min(sK[4 * i + 2], sK[4 * i + 3]));
        const int t = i / H, x = i - t * H;
        if (x < g.w) {
            const size_t o = ((size_t)pair * g.h + y0 + t) * g.w + x;
            web[o] = (i32)(key & 511u) + 1;
            if (best) best[o] = (i32)(key >> 9);
        }
    }
}

template <int H>
static const void *strip_ptr(bool ssd, int ns)
{
    if (ssd) {
        if constexpr (H <= 5) return ns == 1 ? (const void *)k_cost_strip<H, true, 1> : (const void *)k_cost_strip<H, true, 2>;
        else return nullptr;
    }
    return ns == 1 ? (const void *)k_cost_strip<H, false, 1> : (const void *)k_cost_strip<H, false, 2>;
}

// the strip of the fast kernels' launches; returns -1 if this shape is not built (the caller runs the general kernel on it)
int sm_cost_strip_launch(const sm_plan *plan, const uint8_t *d_left, const uint8_t *d_right, int cost, int pairs,
                         int32_t *d_web, int32_t *d_best, hipStream_t stream)
{
    const int half = plan->square_width / 2;
    const bool ssd = cost == SM_COST_SSD;
    StripGeom g;
    g.w = plan->width; g.h = plan->height; g.D = plan->num_shifts;
    if (half < 1 || half > 10 || g.D > 512 || (ssd && half > 5) || plan->opt.cost_kernel == 3) return -1;
    const int ns = (g.D + 255) / 256;
    const int nwd = (2 * half + 3) / 4;
    // a thread reads dwords (d >> 2) .. (d >> 2) + nwd of a right row, d < 256 ns
    g.rw = 64 * ns + nwd + 1;
    // about one workgroup per CU: the launch's duration is one workgroup's
    int th = (int)(((long long)g.h * pairs + 255) / 256);
    th = th < 4 ? 4 : th > 32 ? 32 : th;
    th = th < g.h ? th : g.h;
    g.tile_h = th;
    g.nsr = th + 2 * half;
    const size_t lds = 4 * ((size_t)g.nsr * (nwd + g.rw) + (size_t)th * half * 4);
    if (lds > 64 * 1024) return -1;
    const void *fn = nullptr;
    switch (half) {
    case 1: fn = strip_ptr<1>(ssd, ns); break;
    case 2: fn = strip_ptr<2>(ssd, ns); break;
    case 3: fn = strip_ptr<3>(ssd, ns); break;
    case 4: fn = strip_ptr<4>(ssd, ns); break;
    case 5: fn = strip_ptr<5>(ssd, ns); break;
    case 6: fn = strip_ptr<6>(ssd, ns); break;
    case 7: fn = strip_ptr<7>(ssd, ns); break;
    case 8: fn = strip_ptr<8>(ssd, ns); break;
    case 9: fn = strip_ptr<9>(ssd, ns); break;
    case 10: fn = strip_ptr<10>(ssd, ns); break;
    }
    if (!fn) return -1;
    void *args[] = {(void *)&d_left, (void *)&d_right, (void *)&d_web, (void *)&d_best, (void *)&g};
    const hipError_t e = hipLaunchKernel(fn, dim3(1, (g.h + th - 1) / th, pairs), dim3(256), args, lds, stream);
    if (e != hipSuccess) return sm_fail(SM_ERR_HIP, "sm_cost_wta: %s", hipGetErrorString(e));
    return SM_OK;
}
