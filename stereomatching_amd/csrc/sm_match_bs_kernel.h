// sm_match_bs_kernel.h -- the hot path, BIT-SLICED: match cost -> S x S window sum
// -> masked score -> winner-take-all, 32 pixels per lane operation.
//
// Same contract as sm_match.hip (which remains the general kernel for window
// sizes not instantiated here); same packed ext image as input.
//
// Why: on gfx950 only v_and/or/xor/add/sub/lshr/ashr/bitop3 issue at full
// rate; v_bcnt, v_bfe, v_max, v_lshl*, v_cmp ... take twice as long
// (tools/ubench_valu*.hip).  The popcount/max kernel is bound
// by exactly those.  Here every number lives as BIT PLANES: plane k of a value
// is a 32-bit word holding bit k of that value for 32 neighbouring pixels, and
// all arithmetic is built from v_bitop3 / v_and / v_xor (full rate, 32 pixels
// at a time):
//   x_i   = L(x+i) ^ R(x+i+d)                      mismatch bit of window column i
//   Hx    = sum_i x_i        carry-save adder tree  (N inputs -> HB planes)
//   Sx   += Hx(new row) - Hx(old row)               signed difference, one ripple add on SB planes
//   upd   = centre_match & (Sx <= B)                borrow chain of B - Sx
//   B     = upd ? Sx : B ;  arg = upd ? d : arg     v_bitop3 selects
// Scores are kept as MISMATCH counts: the reference's score is
// (#valid window taps) - Sx and the tap count does not depend on the shift, so
// "highest score, last shift wins" == "lowest Sx, last shift wins" (ascending
// d with <=).  best = taps - B is formed when the planes are turned back into
// integers.  B starts at all ones (2^SB - 1 > N*N), which doubles as the "no
// shift matched" marker (-> web = D, best = 0; src/stereo.c:211-218).
//
// Lane = one 32-pixel word x DS = 16 shifts, marching down the tile with the
// 16 x SB sum planes in VGPRs.  The shift range of a word is split over
// nl = D/16 adjacent lanes, merged per row with DPP row operations on the
// planes (lexicographic: lower Sx, then the lane holding the higher shifts).
// After the merge each of the nl lanes turns 32/nl pixels back into integers
// and stores them.

// This header is the kernel template; its instantiations are spread over four translation
// units that compile side by side (sm_match_bs.hip: one wave per workgroup, 16 shifts per
// lane, and the host side; sm_match_bs_ds8.hip: 8 shifts per lane; sm_match_bs_duo.hip /
// sm_match_bs_duo8.hip: two-wave workgroups).  Each defines SM_BS_TU, a name of its own.
#pragma once
#include "sm_internal.h"

#ifndef SM_BS_TU
#error "define SM_BS_TU (a token naming the translation unit) before including this header"
#endif
#define SM_BS_CAT2(a, b) a##b
#define SM_BS_CAT(a, b) SM_BS_CAT2(a, b)

#ifndef SM_BS_WAVES
#define SM_BS_WAVES 2   // min waves per SIMD: keeps VGPR + AGPR <= 256 (one AGPR more halves the occupancy)
#endif

// Diagnostic build only (-DSM_STAMPS, tools/wave_timeline.py): every wave records
// when it started, finished staging, finished its warm-up rows and ended (constant
// 100 MHz s_memrealtime and shader-clock s_memtime) plus where it ran (HW_ID,
// XCC_ID), into a buffer of its own that nothing else reads.  The product library
// is built without it and contains none of this.
#ifdef SM_STAMPS
// one pointer per translation unit (each is a code object of its own); sm_debug_set_stamps
// in sm_match_bs.hip sets them all
static __device__ unsigned long long *g_sm_stamps;
int SM_BS_CAT(sm_bs_set_stamps_, SM_BS_TU)(void *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sm_stamps), &buf, sizeof buf) == hipSuccess ? 0 : SM_ERR_HIP;
}
#define SM_STAMP(slot)                                                                          \
    do {                                                                                        \
        if ((threadIdx.x & 63) == 0 && g_sm_stamps) {        /* one record per WAVE */           \
            const size_t wg_ = (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) \
                               * (blockDim.x >> 6) + (threadIdx.x >> 6);                        \
            g_sm_stamps[wg_ * 10 + 2 * (slot)] = __builtin_amdgcn_s_memrealtime();              \
            g_sm_stamps[wg_ * 10 + 2 * (slot) + 1] = __builtin_amdgcn_s_memtime();              \
            if ((slot) == 0)                                                                    \
                g_sm_stamps[wg_ * 10 + 8] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | \
                                            (unsigned)__builtin_amdgcn_s_getreg(63492);         \
        }                                                                                       \
    } while (0)
#else
#define SM_STAMP(slot) do { } while (0)
#endif

template <int IMM>
__device__ __forceinline__ u32 bop(u32 a, u32 b, u32 c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, IMM);   // bit (a<<2|b<<1|c) of IMM
}
#define BOP_XOR3 0x96      // a ^ b ^ c
#define BOP_MAJ 0xE8       // majority(a, b, c)
#define BOP_BORROW 0x8E    // majority(~a, b, c): borrow out of a - b - c
#define BOP_SEL 0xCA       // a ? b : c
#define BOP_XOR_AND 0x28   // (a ^ b) & c
#define BOP_UPD 0x41       // ~(a ^ b) & ~c

__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// ---------------------------------------------------------------------------
// LDS reads that stay in flight (SM_BS_PREFETCH).  Left to itself the compiler issues a
// row's ds_reads right in front of their first use and the wave then sits in s_waitcnt
// for the LDS latency, twice per output row, with one other wave on the SIMD to cover
// for it.  These reads are issued one whole row EARLIER: VOLATILE loads, which the
// scheduler may not reorder among themselves or move across the scheduling barrier that
// follows them, so they stay where they are written; the compiler still tracks them and
// places the s_waitcnt in front of the first use of each value.
//
// (Until round 2 these were inline-asm ds_reads with a hand-placed s_waitcnt.  That is
// only correct while the register allocator leaves the destination registers alone until
// the wait: the two largest windows, 19 x 19 and 21 x 21 with the ghost border, need more
// registers than a wave has, the allocator spilled some of the prefetched values straight
// after the asm statement -- before the data had arrived -- and reloaded garbage: wrong
// sums from the third row of a tile on, found by tests/test_hip_gpu.py::
// test_every_built_kernel_matches_oracle.  A volatile load is an ordinary load to the
// allocator and the wait-count pass, so a spill waits for the data like any other use.
// Same speed: C3 93.2 vs 93.0 us, C5 193.3 vs 193.3, 21 x 21 at 4K 78.5 vs 78.8, same
// device, profiles/r02/ab_prefetch_volatile.txt.)
// ---------------------------------------------------------------------------
#ifndef SM_BS_PREFETCH
#define SM_BS_PREFETCH 1
#endif
// the shift lanes of a word merged through LDS every four rows (g.xmerge) instead of per row with DPP
#ifndef SM_BS_XMERGE
#define SM_BS_XMERGE 1
#endif
typedef unsigned long long u64;
struct RawRow { u64 l01; u32 l2; u64 r01, r23; };      // 3 left words, 4 right words
struct RawCentre { u32 l; u64 r01; u32 r2; };          // centre word, 3 right words
// 32-bit LDS addresses; the 8-byte loads are only 4-byte aligned (-> ds_read2_b32)
typedef __attribute__((address_space(3))) const volatile u32 lds_vu32;
typedef u64 __attribute__((aligned(4))) u64a4;
typedef __attribute__((address_space(3))) const volatile u64a4 lds_vu64;
__device__ __forceinline__ void lds_issue(RawRow &o, u32 aL, u32 aR)
{
    o.l01 = *(lds_vu64 *)(uintptr_t)aL;
    o.l2 = *(lds_vu32 *)(uintptr_t)(aL + 8);
    o.r01 = *(lds_vu64 *)(uintptr_t)aR;
    o.r23 = *(lds_vu64 *)(uintptr_t)(aR + 8);
}
__device__ __forceinline__ void lds_issue(RawCentre &o, u32 aL, u32 aR)
{
    o.l = *(lds_vu32 *)(uintptr_t)aL;
    o.r01 = *(lds_vu64 *)(uintptr_t)aR;
    o.r2 = *(lds_vu32 *)(uintptr_t)(aR + 8);
}

// result stores: 0 = nt (streaming hint, the line still stays in L2), 1 = sc1
// (write-through: the bytes leave L2 while the kernel runs instead of in one
// write-back burst when it ends), 2 = plain
#ifndef SM_BS_STORE
#define SM_BS_STORE 1   // measured, same device: sc1 1 % faster than nt at C3 / C4 x 8, 4 % at C2
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_map4(i32 *p, v4i v)
{
#if SM_BS_STORE == 1
    // (the s_nop belongs to the statement: the compiler does not see a store of more than 8 bytes
    // here, so nothing else keeps a VALU write of the data registers one wait state away from it)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#elif SM_BS_STORE == 2
    *reinterpret_cast<v4i *>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<v4i *>(p));
#endif
}

constexpr int bits_for(int v) { int b = 0; while ((1 << b) <= v) b++; return b; }   // v < 2^b

// ---------------------------------------------------------------------------
// carry-save tree: N one-bit inputs -> their count on HB planes.  Everything
// about the wiring is known at compile time; after unrolling only the
// v_bitop3 / v_xor / v_and of the adders remain (5 full + 2 half adders for 9).
// ---------------------------------------------------------------------------
template <int N, int HB>
__device__ __forceinline__ void count_bits(const u32 (&x)[N], u32 (&h)[HB])
{
    u32 q[2 * N + 2];          // wires of the current weight, used as a queue
    int head = 0, tail = 0;
#pragma unroll
    for (int i = 0; i < N; i++) q[tail++] = x[i];
#pragma unroll
    for (int w = 0; w < HB; w++) {
        u32 nx[N + 1];         // carries: wires of the next weight
        int nn = 0;
#pragma unroll
        for (int it = 0; it < N; it++) {
            if (tail - head >= 3) {
                const u32 a = q[head], b = q[head + 1], c = q[head + 2];
                head += 3;
                q[tail++] = bop<BOP_XOR3>(a, b, c);
                nx[nn++] = bop<BOP_MAJ>(a, b, c);
            }
        }
        if (tail - head == 2) {
            const u32 a = q[head], b = q[head + 1];
            head += 2;
            q[tail++] = a ^ b;
            nx[nn++] = a & b;
        }
        h[w] = tail - head == 1 ? q[head] : 0u;
        head = tail = 0;
#pragma unroll
        for (int i = 0; i < N + 1; i++)
            if (i < nn) q[tail++] = nx[i];
    }
}

// s += h  (s: SB planes, h: HB planes, HB <= SB; the sum is known to fit)
template <int SB, int HB>
__device__ __forceinline__ void add_planes(u32 (&s)[SB], const u32 (&h)[HB])
{
    u32 c = s[0] & h[0];
    s[0] ^= h[0];
#pragma unroll
    for (int k = 1; k < SB; k++) {
        // carry first, then the plane in place: a sum plane stays in its register
        if (k < HB) {
            const u32 cn = bop<BOP_MAJ>(s[k], h[k], c);
            s[k] = bop<BOP_XOR3>(s[k], h[k], c);
            c = cn;
        } else {
            const u32 cn = k + 1 < SB ? s[k] & c : 0u;
            s[k] ^= c;
            c = cn;
        }
    }
}

// s += hn - ho in one pass: the HB-plane difference in two's complement (its sign is
// the borrow out), then one ripple add of the sign-extended difference
template <int SB, int HB>
__device__ __forceinline__ void addsub_planes(u32 (&s)[SB], const u32 (&hn)[HB], const u32 (&ho)[HB])
{
    u32 dl[HB];
    u32 b = ~hn[0] & ho[0];
    dl[0] = hn[0] ^ ho[0];
#pragma unroll
    for (int k = 1; k < HB; k++) {
        dl[k] = bop<BOP_XOR3>(hn[k], ho[k], b);
        b = bop<BOP_BORROW>(hn[k], ho[k], b);
    }
    const u32 sg = b;                       // all higher planes of the difference
    u32 c = s[0] & dl[0];
    s[0] ^= dl[0];
#pragma unroll
    for (int k = 1; k < SB; k++) {
        const u32 a = k < HB ? dl[k] : sg;
        const u32 cn = k + 1 < SB ? bop<BOP_MAJ>(s[k], a, c) : 0u;
        s[k] = bop<BOP_XOR3>(s[k], a, c);
        c = cn;
    }
}

// ---------------------------------------------------------------------------
// LOCKSTEP forms (SM_BS_LOCKSTEP).  Measured on gfx950 with exactly two waves per SIMD
// (tools/ubench_issue.hip, tools/ubench_body.hip): a VALU instruction that reads the
// result of the instruction issued 1 / 2 / 3 instructions earlier in its own wave costs
// the SIMD 8 / 4 / ~2.7 issue cycles instead of 2, and the OLDER wave of the pair keeps
// its full rate while the younger one is left with what remains (in the kernel: older
// waves done after 74 us, younger after 107).  From a distance of 4 on the SIMD issues
// one instruction every 2 cycles.  The compiler models VALU latency as 1 and lines
// dependent instructions up back to back (13-17 % of this kernel's instructions had
// distance 1, another 20-36 % distance 2).  So the arithmetic below is written as IT
// independent items advanced side by side, operation by operation -- every dependency
// then has a distance >= IT -- and each operation is followed by a scheduling barrier
// (SM_PIN) so that the order written here is the order issued.
// ---------------------------------------------------------------------------
#ifndef SM_BS_LOCKSTEP
#define SM_BS_LOCKSTEP 1
#endif
#define SM_PIN() __builtin_amdgcn_sched_barrier(0)
// Re-phase the two waves of a SIMD.  Measured (tools/gen_ubench_bankrules.py, two waves per
// SIMD): ONE half-rate VALU instruction (v_alignbit, DPP moves, v_bfe, v_perm, v_lshl_or,
// v_and_or; v_lshlrev / v_mul_u32_u24 half as bad) leaves the pair in a state where the
// older wave issues every 4 cycles and the younger only every 8 -- 2.67 cycles per
// instruction for the SIMD instead of 2 -- and they STAY there until a scalar instruction
// passes: 1 alignbit per 64 v_bitop3 costs 3.9 cycles per instruction, the same with an
// s_nop behind each alignbit 2.7.  SM_SYNC(level) emits that s_nop where SM_BS_NOP >= level.
#ifndef SM_BS_PRIO
#define SM_BS_PRIO 0      // a STATIC priority only swaps which wave of the pair is starved (measured)
#endif
#ifndef SM_BS_SLICE
#define SM_BS_SLICE 16    // swap every 65536 cycles (~31 us): 13 / 15 / 17 measured slower
#endif
#ifndef SM_BS_PATTERN
#define SM_BS_PATTERN 0xF0F0F0F0u   // 4 units (65536 cycles) per slice; 2 / 3 / 5 / 6 units, shifted phases, unequal duty
                                    // cycles and the inverse measured, for both workgroup shapes: all within +-1.5 %
#endif
#ifndef SM_BS_NOP
#define SM_BS_NOP 3
#endif
#define SM_SYNC(level) do { if (SM_BS_NOP >= (level)) { asm volatile("s_nop 0"); SM_PIN(); } } while (0)
#define BOP_ANDN 0x0C      // ~a & b
#define BOP_ORN 0xCF       // ~a | b   (a ? b : all ones)
#define BOP_XNOR 0xC3      // ~(a ^ b)            (c ignored)

// IT counters side by side: inputs come from xin(item, i), made when first used
template <int N, int HB, int IT, typename XF>
__device__ __forceinline__ void count_lockstep(XF xin, u32 (&h)[IT][HB])
{
    u32 q[IT][2 * N + 2];      // wires of the current weight; [0, N) of weight 0 are the inputs
    int head = 0, tail = N;
#pragma unroll
    for (int w = 0; w < HB; w++) {
        u32 nx[IT][N + 1];     // carries: wires of the next weight
        int nn = 0;
#pragma unroll
        for (int adder = 0; adder < N; adder++) {
            if (tail - head >= 3) {
                u32 a[IT], b[IT], c[IT];
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const bool in0 = w == 0 && head < N, in1 = w == 0 && head + 1 < N, in2 = w == 0 && head + 2 < N;
                    a[it] = in0 ? xin(it, in0 ? head : 0) : q[it][head];
                    b[it] = in1 ? xin(it, in1 ? head + 1 : 0) : q[it][head + 1];
                    c[it] = in2 ? xin(it, in2 ? head + 2 : 0) : q[it][head + 2];
                }
#pragma unroll
                for (int it = 0; it < IT; it++) { q[it][tail] = bop<BOP_XOR3>(a[it], b[it], c[it]); SM_PIN(); }
#pragma unroll
                for (int it = 0; it < IT; it++) { nx[it][nn] = bop<BOP_MAJ>(a[it], b[it], c[it]); SM_PIN(); }
                SM_SYNC(2);
                head += 3; tail++; nn++;
            }
        }
        if (tail - head == 2) {
            u32 a[IT], b[IT];
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const bool in0 = w == 0 && head < N, in1 = w == 0 && head + 1 < N;
                a[it] = in0 ? xin(it, in0 ? head : 0) : q[it][head];
                b[it] = in1 ? xin(it, in1 ? head + 1 : 0) : q[it][head + 1];
            }
#pragma unroll
            for (int it = 0; it < IT; it++) { q[it][tail] = a[it] ^ b[it]; SM_PIN(); }
#pragma unroll
            for (int it = 0; it < IT; it++) { nx[it][nn] = a[it] & b[it]; SM_PIN(); }
            head += 2; tail++; nn++;
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const bool in0 = w == 0 && head < N;
            h[it][w] = tail - head == 1 ? (in0 ? xin(it, in0 ? head : 0) : q[it][head]) : 0u;
        }
        head = 0; tail = 0;
#pragma unroll
        for (int i = 0; i < N + 1; i++)
            if (i < nn) {
#pragma unroll
                for (int it = 0; it < IT; it++) q[it][tail] = nx[it][i];
                tail++;
            }
    }
}

// S[dd0 + g] += h[g] for g < GS side by side (warm-up rows)
template <int SB, int HB, int GS, int DS>
__device__ __forceinline__ void add_lockstep(u32 (&S)[DS][SB], int dd0, const u32 (&h)[GS][HB])
{
    u32 c[GS];
#pragma unroll
    for (int g = 0; g < GS; g++) { c[g] = S[dd0 + g][0] & h[g][0]; SM_PIN(); }
#pragma unroll
    for (int g = 0; g < GS; g++) { S[dd0 + g][0] ^= h[g][0]; SM_PIN(); }
#pragma unroll
    for (int k = 1; k < SB; k++) {
        u32 cn[GS];
#pragma unroll
        for (int g = 0; g < GS; g++) {
            if (k + 1 < SB) { cn[g] = k < HB ? bop<BOP_MAJ>(S[dd0 + g][k], h[g][k], c[g]) : (S[dd0 + g][k] & c[g]); SM_PIN(); }
            else cn[g] = 0;
        }
#pragma unroll
        for (int g = 0; g < GS; g++) {
            if (k < HB) S[dd0 + g][k] = bop<BOP_XOR3>(S[dd0 + g][k], h[g][k], c[g]);
            else S[dd0 + g][k] ^= c[g];
            SM_PIN();
        }
#pragma unroll
        for (int g = 0; g < GS; g++) c[g] = cn[g];
        SM_SYNC(3);
    }
}

// S[dd0 + g] += hn[g] - ho[g] for g < GS side by side: the HB-plane differences in two's
// complement (sign = borrow out), then one ripple add of the sign-extended differences
template <int SB, int HB, int GS, int DS>
__device__ __forceinline__ void addsub_lockstep(u32 (&S)[DS][SB], int dd0, const u32 (&hn)[GS][HB],
                                                const u32 (&ho)[GS][HB])
{
    u32 dl[GS][HB], b[GS];
#pragma unroll
    for (int g = 0; g < GS; g++) { b[g] = bop<BOP_ANDN>(hn[g][0], ho[g][0], 0u); SM_PIN(); }
#pragma unroll
    for (int g = 0; g < GS; g++) { dl[g][0] = hn[g][0] ^ ho[g][0]; SM_PIN(); }
#pragma unroll
    for (int k = 1; k < HB; k++) {
#pragma unroll
        for (int g = 0; g < GS; g++) { dl[g][k] = bop<BOP_XOR3>(hn[g][k], ho[g][k], b[g]); SM_PIN(); }
#pragma unroll
        for (int g = 0; g < GS; g++) { b[g] = bop<BOP_BORROW>(hn[g][k], ho[g][k], b[g]); SM_PIN(); }
        SM_SYNC(3);
    }
    u32 c[GS];
#pragma unroll
    for (int g = 0; g < GS; g++) { c[g] = S[dd0 + g][0] & dl[g][0]; SM_PIN(); }
#pragma unroll
    for (int g = 0; g < GS; g++) { S[dd0 + g][0] ^= dl[g][0]; SM_PIN(); }
#pragma unroll
    for (int k = 1; k < SB; k++) {
        u32 cn[GS];
#pragma unroll
        for (int g = 0; g < GS; g++) {
            const u32 a = k < HB ? dl[g][k] : b[g];
            if (k + 1 < SB) { cn[g] = bop<BOP_MAJ>(S[dd0 + g][k], a, c[g]); SM_PIN(); } else cn[g] = 0;
        }
#pragma unroll
        for (int g = 0; g < GS; g++) {
            const u32 a = k < HB ? dl[g][k] : b[g];
            S[dd0 + g][k] = bop<BOP_XOR3>(S[dd0 + g][k], a, c[g]); SM_PIN();
        }
#pragma unroll
        for (int g = 0; g < GS; g++) c[g] = cn[g];
        SM_SYNC(3);
    }
}

template <int V> struct IntTag { static constexpr int value = V; };

template <int CTRL>
__device__ __forceinline__ u32 dpp(u32 v)
{
    // every lane of these permutations has a valid source: bound_ctrl with a zero `old`
    // lets the compiler emit ONE v_mov_b32_dpp (tying `old` to v costs a copy first)
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// partner lane of merge step K (lane ^ (1 << K)) for K = 0..3 via DPP, 4/5 via
// the LDS crossbar
template <int K>
__device__ __forceinline__ u32 from_partner(u32 v)
{
    if (K == 0) return dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    if (K == 1) return dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    if (K == 2) return dpp<0x141>(v);           // row_half_mirror: i <-> 7 - i
    if (K == 3) return dpp<0x140>(v);           // row_mirror:      i <-> 15 - i
    return (u32)__shfl_xor((int)v, 1 << K);
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------

// CAP2: claim a register beyond the 168 a wave may hold when three share a SIMD, so
// that the hardware admits at most TWO waves of this kernel per SIMD.  For a grid
// that fits the chip at two waves per SIMD this is what keeps the waves evenly
// spread: measured (8 x 1080p, 7 x 7, 159 VGPRs), a launch that follows a different
// kernel finds the SIMDs empty and the dispatcher stacks three waves on some of them
// while others get one -- 117 us instead of 89; behind a launch of the same kernel
// the waves inherit the previous, even placement.  A per-CU cap (the LDS request in
// sm_match_configure) cannot prevent it, a per-SIMD one does.
//
// DUO: workgroups of TWO waves that share their warm-up.  The workgroup owns 2 * tile_h
// rows; wave 1 starts at the middle row m and slides DOWN, wave 0 starts at row m - 1 and
// slides UP.  Their first windows, rows m - HALF .. m + HALF and m - 1 - HALF .. m - 1 + HALF,
// have N - 1 rows in common: each wave counts HALF of those, the partial sums are swapped
// through LDS, and each adds the one row that is its own -- HALF + 1 warm-up rows per wave
// instead of N.
template <int N, int DS, bool FULLD, bool GHOST, bool CAP2 = false, bool DUO = false>
__global__ __launch_bounds__(DUO ? 128 : 64, SM_BS_WAVES) void k_match_bs(const u32 *__restrict__ ext,
                                                 i32 *__restrict__ web, i32 *__restrict__ best,
                                                 const MatchGeom g)
{
    constexpr int HALF = N / 2;
    constexpr int HB = bits_for(N);             // planes of a horizontal count
    constexpr int SB = bits_for(N * N);         // planes of a window count; 2^SB - 1 > N*N
    constexpr int AB = DS == 16 ? 4 : DS == 8 ? 3 : 2;   // planes of the in-lane shift index
    constexpr int ABMAX = AB + 6;               // after merging up to 64 lanes
    static_assert(SB <= 16, "window counts are moved through two byte lanes");
    static_assert((1 << SB) - 1 > N * N, "the all-ones marker must not be a real count");
    static_assert(N + DS - 1 + 31 < 96, "right window must fit three words");

    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    if (CAP2) asm volatile("" ::: "v175");
    if (web == nullptr) return;     // the plan's set-up launch: loads the code object, does nothing
    const int tid = DUO ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;     // lane of the wave
    const int wv = DUO ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int pair = blockIdx.z;
    SM_STAMP(0);
#if SM_BS_PRIO
    // Unequal priorities for the waves that share a SIMD.  With equal priority the SIMD's
    // arbiter serves the OLDEST wave first, and once any half-rate instruction (v_alignbit,
    // DPP, v_bfe, ...) has passed, the younger wave gets an issue slot only every 8 cycles
    // while the older keeps its 4 (tools/gen_ubench_bankrules.py, "mix 16": 4.0 cycles per
    // instruction for the SIMD; with a raised priority on the odd hardware wave slots 2.5;
    // this kernel: older waves done after 74 us, younger after 107).  The hardware wave
    // slot (HW_ID bits 3:0) tells the co-resident waves apart: slot parity -> priority.
    if (__builtin_amdgcn_s_getreg(63492) & 1) __builtin_amdgcn_s_setprio(SM_BS_PRIO);
#endif
#if SM_BS_SLICE
    // Time-sliced priority: the wave whose hardware slot parity equals bit SM_BS_SLICE of
    // the shader clock runs at raised priority, the other at 0, and the roles swap every
    // 2^SM_BS_SLICE cycles (with SM_BS_PATTERN: after a schedule counted from the wave's own
    // start) -- both read the same clock, so exactly one of a pair is favoured at any time.
    // (A feedback variant -- every wave publishes its finished rows per SIMD and slot in
    // global memory, the one that is behind takes the priority -- was built and measured
    // slower, 96.6 vs 91.3 us: a store, a load and a v_readfirstlane per row, and the waves
    // still finished 15 us apart.)  The SIMD serves its favoured wave at the rate of a wave alone
    // and gives the other what is left (measured: 4.9 vs 9.0 cycles per instruction here);
    // without the swap the favoured wave finishes a third earlier and the SIMD then runs
    // half empty until the other is done.
    // Which of a SIMD's two waves a slice favours: bit g.prio_shift of HW_ID -- bit 0 of the wave slot, or
    // (two-wave workgroups of a launch that fits the chip in one round) bit 0 of TG_ID, the workgroup's slot
    // on its CU.  The two waves of a SIMD differ in either (tools/wave_timeline.py: 1016 of 1016 pairs), but
    // the two waves of a WORKGROUP share the second: their barrier-coupled warm-up is then favoured as a
    // whole instead of running at the pace of whichever of the two is not (a workgroup with one wave in
    // each slot parity -- half of them -- was never favoured as a whole).  In a launch of several rounds
    // later workgroups land in whatever slot is free and the bit no longer separates a SIMD's pair:
    // those keep the wave slot.
    const unsigned slot_parity = (__builtin_amdgcn_s_getreg(63492) >> g.prio_shift) & 1;
    // the clock is read one row ahead of its use (s_memtime is a scalar memory read: its
    // value takes ~100 cycles to arrive, and a wave that uses it at once waits that long)
    unsigned long long clk = __builtin_amdgcn_s_memtime();
#if SM_BS_PATTERN
    // slices counted from the wave's own start (the waves of a launch start within 0.5 us of
    // each other): bit k of the plan's pattern (g.prio_pattern: SM_BS_PATTERN unless SM_PATTERN
    // in the environment overrides it for tuning) says which slot parity is favoured during
    // the k-th unit of 2^g.prio_unit cycles (16384, ~8 us; short launches take finer ones), so the schedule is the
    // same in every launch
    const unsigned long long clk0 = clk;
    // (g.prio_on_change, tuning: s_setprio only when the wanted priority changes instead of once per row --
    // on part of the pool a wave that re-issues it every row is served as if it had none)
    unsigned prio_now = 2;
#define SM_SLICE_PRIO()                                                               \
    do {                                                                              \
        const unsigned unit_ = (unsigned)((clk - clk0) >> g.prio_unit) & 31u;         \
        const unsigned want_ = ((g.prio_pattern >> unit_) ^ slot_parity) & 1;         \
        if (!g.prio_on_change || want_ != prio_now) {                                 \
            if (want_) __builtin_amdgcn_s_setprio(3);                                 \
            else __builtin_amdgcn_s_setprio(0);                                       \
            prio_now = want_;                                                         \
        }                                                                             \
    } while (0)
#else
#define SM_SLICE_PRIO()                                                               \
    do {                                                                              \
        if ((((unsigned)(clk >> SM_BS_SLICE)) ^ slot_parity) & 1) __builtin_amdgcn_s_setprio(3); \
        else __builtin_amdgcn_s_setprio(0);                                           \
    } while (0)
#endif
#define SM_SLICE_READ() do { clk = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SM_SLICE_PRIO() do { } while (0)
#define SM_SLICE_READ() do { } while (0)
#endif
    int tile_x, tile_y;
    sm_xcd_tile(g.tiles_x, g.tiles_y, tile_x, tile_y);
    const int tx0 = tile_x * g.tw;
    const int ty0 = tile_y * (DUO ? 2 * g.tile_h : g.tile_h);
    const int plw = g.plw, prw = g.prw, nsr = g.nsr;
    const u32 *extL = ext + (size_t)pair * 2 * g.ext_image_words;
    const u32 *extR = extL + g.ext_image_words;
    u32 *pL = lds;                  // [nsr][plw]
    u32 *pR = pL + nsr * plw;       // [nsr][prw]

    // ---- stage the tile's rows (+ window halo): coalesced dword row loads
    {
        const int wx0 = tx0 >> 5;
        const int per_row = plw + prw;
        for (int it = threadIdx.x; it < nsr * per_row; it += DUO ? 128 : 64) {
            const int row = it / per_row, k = it - row * per_row;
            const bool is_r = k >= plw;
            const int kk = is_r ? k - plw : k;
            const u32 v = (is_r ? extR : extL)[(size_t)(ty0 + row) * g.ext_words + wx0 + kk];
            if (is_r) pR[row * prw + kk] = v; else pL[row * plw + kk] = v;
        }
    }
    __syncthreads();
    SM_STAMP(1);

    // ---- lane role
    const int s = tid & (g.nl - 1);           // which 16 shifts
    const int wi = tid >> g.log2nl;           // which 32-pixel word of the tile
    const int d0 = s * DS;
    const int x0 = tx0 + 32 * wi;
    // bit offsets in a staged row (bit = SM_PADT + tile-local x)
    const int bL = SM_PADT + 32 * wi - HALF, bR = bL + d0;
    const int wL = bL >> 5, shL = bL & 31, wR = bR >> 5, shR = bR & 31;
    const int bRc = SM_PADT + 32 * wi + d0;
    const int wRc = bRc >> 5, shRc = bRc & 31;
    const int wLc = 1 + wi;

    // ghost: validity of the window columns (bit i of cv = column x0 - HALF + i)
    u32 cvv[N];
    if (GHOST) {
        // bits [lo, hi) of the 64-bit mask {c1, c0}: the columns x0 - HALF + i inside [0, w)
        const int lo = min(64, max(0, HALF - x0)), hi = min(64, max(0, g.w + HALF - x0));
        const unsigned long long below_hi = hi >= 64 ? ~0ull : (1ull << hi) - 1ull;
        const unsigned long long below_lo = lo >= 64 ? ~0ull : (1ull << lo) - 1ull;
        const unsigned long long cm = hi > lo ? below_hi & ~below_lo : 0ull;
        const u32 c0 = (u32)cm, c1 = (u32)(cm >> 32);
#pragma unroll
        for (int i = 0; i < N; i++) cvv[i] = i ? alignbit(c1, c0, i) : c0;
    }
    // shifts >= D of this lane never match
    u32 dvalid = (1u << DS) - 1u;
    if (!FULLD) {
        const int dlim = g.D - d0;
        dvalid = dlim >= DS ? (1u << DS) - 1u : (dlim <= 0 ? 0u : ((1u << dlim) - 1u));
    }

    u32 S[DS][SB];
#pragma unroll
    for (int dd = 0; dd < DS; dd++)
#pragma unroll
        for (int k = 0; k < SB; k++) S[dd][k] = 0;

    // the views one window row contributes: N pre-shifted words of the left row and
    // the three words the right row's sliding views are cut from
    struct RowViews { u32 lv[N]; u32 rw[3]; };
    auto load_views = [&](int srow, RowViews &v) {
        const u32 *rl = pL + srow * plw + wL;
        const u32 *rr = pR + srow * prw + wR;
        const u32 l0 = alignbit(rl[1], rl[0], shL), l1 = alignbit(rl[2], rl[1], shL);
#pragma unroll
        for (int k = 0; k < 3; k++) v.rw[k] = alignbit(rr[k + 1], rr[k], shR);
#pragma unroll
        for (int i = 0; i < N; i++) v.lv[i] = i ? alignbit(l1, l0, i) : l0;
    };
    auto views_of = [&](const RawRow &q, RowViews &v) {
        const u32 a0 = (u32)q.l01, a1 = (u32)(q.l01 >> 32), a2 = q.l2;
        const u32 r0 = (u32)q.r01, r1 = (u32)(q.r01 >> 32), r2 = (u32)q.r23, r3 = (u32)(q.r23 >> 32);
        const u32 l0 = alignbit(a1, a0, shL), l1 = alignbit(a2, a1, shL);
        v.rw[0] = alignbit(r1, r0, shR); v.rw[1] = alignbit(r2, r1, shR); v.rw[2] = alignbit(r3, r2, shR);
#pragma unroll
        for (int i = 0; i < N; i++) v.lv[i] = i ? alignbit(l1, l0, i) : l0;
    };
    auto rview = [&](const RowViews &v, int j) -> u32 {
        return (j & 31) ? alignbit(v.rw[(j >> 5) + 1], v.rw[j >> 5], j & 31) : v.rw[j >> 5];
    };
#if !SM_BS_LOCKSTEP
    // mismatch count of shift dd in one row (win = the N right views of this shift)
    auto count_row = [&](const RowViews &v, const u32 (&win)[N], u32 (&h)[HB]) {
        u32 x[N];
#pragma unroll
        for (int i = 0; i < N; i++)
            x[i] = GHOST ? bop<BOP_XOR_AND>(v.lv[i], win[i], cvv[i]) : (v.lv[i] ^ win[i]);
        count_bits<N, HB>(x, h);
    };
#endif

#if SM_BS_LOCKSTEP
    constexpr int GW = DS >= 4 ? 4 : DS;      // shifts side by side in a warm-up row
    constexpr int GS = 2;                     // ... in a steady-state row (2 x {row in, row out})
    static_assert(DS % GW == 0 && DS % GS == 0, "shift groups");
    // warm-up: one window row into all sums, GW shifts side by side
    auto slide_in = [&](int srow) {
        RowViews v;
        load_views(srow, v);
        u32 rv[N + GW - 1];                   // right views dd0 ... dd0 + GW + N - 2 of the group
#pragma unroll
        for (int m = 0; m < N - 1; m++) rv[m + GW] = rview(v, m);
#pragma unroll
        for (int dd0 = 0; dd0 < DS; dd0 += GW) {
#pragma unroll
            for (int m = 0; m < N - 1; m++) rv[m] = rv[m + GW];
#pragma unroll
            for (int m = 0; m < GW; m++) { rv[N - 1 + m] = rview(v, dd0 + N - 1 + m); SM_PIN(); }
            SM_SYNC(1);
            u32 h[GW][HB];
            count_lockstep<N, HB, GW>([&](int it, int i) -> u32 {
                const u32 x = GHOST ? bop<BOP_XOR_AND>(v.lv[i], rv[it + i], cvv[i]) : (v.lv[i] ^ rv[it + i]);
                SM_PIN();
                return x;
            }, h);
            add_lockstep<SB, HB, GW, DS>(S, dd0, h);
        }
    };
    // steady state: one row in and one row out, GS shifts x {in, out} side by side
    auto slide_views = [&](const RowViews &vn, const RowViews &vo) {
        u32 rn[N + GS - 1], ro[N + GS - 1];
#pragma unroll
        for (int m = 0; m < N - 1; m++) { rn[m + GS] = rview(vn, m); ro[m + GS] = rview(vo, m); }
#pragma unroll
        for (int dd0 = 0; dd0 < DS; dd0 += GS) {
#pragma unroll
            for (int m = 0; m < N - 1; m++) { rn[m] = rn[m + GS]; ro[m] = ro[m + GS]; }
#pragma unroll
            for (int m = 0; m < GS; m++) {
                rn[N - 1 + m] = rview(vn, dd0 + N - 1 + m); SM_PIN();
                ro[N - 1 + m] = rview(vo, dd0 + N - 1 + m); SM_PIN();
            }
            SM_SYNC(1);
            u32 h[2 * GS][HB];                // item 2g = shift dd0 + g row in, 2g + 1 = row out
            count_lockstep<N, HB, 2 * GS>([&](int it, int i) -> u32 {
                const u32 l = (it & 1) ? vo.lv[i] : vn.lv[i];
                const u32 r = (it & 1) ? ro[(it >> 1) + i] : rn[(it >> 1) + i];
                const u32 x = GHOST ? bop<BOP_XOR_AND>(l, r, cvv[i]) : (l ^ r);
                SM_PIN();
                return x;
            }, h);
            u32 hn[GS][HB], ho[GS][HB];
#pragma unroll
            for (int gq = 0; gq < GS; gq++)
#pragma unroll
                for (int k = 0; k < HB; k++) { hn[gq][k] = h[2 * gq][k]; ho[gq][k] = h[2 * gq + 1][k]; }
            addsub_lockstep<SB, HB, GS, DS>(S, dd0, hn, ho);
            SM_SYNC(2);
        }
    };
#else
    // warm-up: one window row into all sums
    auto slide_in = [&](int srow) {
        RowViews v;
        load_views(srow, v);
        u32 win[N];
#pragma unroll
        for (int i = 0; i < N - 1; i++) win[i + 1] = rview(v, i);
#pragma unroll
        for (int dd = 0; dd < DS; dd++) {
#pragma unroll
            for (int i = 0; i < N - 1; i++) win[i] = win[i + 1];
            win[N - 1] = rview(v, dd + N - 1);
            u32 h[HB];
            count_row(v, win, h);
            add_planes<SB, HB>(S[dd], h);
        }
    };
    // steady state: one row in and one row out, applied as a single signed difference
    auto slide_views = [&](const RowViews &vn, const RowViews &vo) {
        u32 wn[N], wo[N];
#pragma unroll
        for (int i = 0; i < N - 1; i++) { wn[i + 1] = rview(vn, i); wo[i + 1] = rview(vo, i); }
#pragma unroll
        for (int dd = 0; dd < DS; dd++) {
#pragma unroll
            for (int i = 0; i < N - 1; i++) { wn[i] = wn[i + 1]; wo[i] = wo[i + 1]; }
            wn[N - 1] = rview(vn, dd + N - 1);
            wo[N - 1] = rview(vo, dd + N - 1);
            u32 hn[HB], ho[HB];
            count_row(vn, wn, hn);
            count_row(vo, wo, ho);
            addsub_planes<SB, HB>(S[dd], hn, ho);
        }
    };
#endif
#if !SM_BS_PREFETCH
    auto slide_both = [&](int srow_new, int srow_old) {
        RowViews vn, vo;
        load_views(srow_new, vn);
        load_views(srow_old, vo);
        slide_views(vn, vo);
    };
#endif

    // Staged row e is image row ty0 - HALF + e.  Ghost rows outside the image need no
    // special case: their ext rows are all zero in BOTH images, so every tap reads
    // 0 ^ 0 = "no mismatch" and the row adds nothing to the sums (the taps that may
    // count are taken care of at the output: best = valid taps - mismatches).
    //
    // Output row t of this wave is image row y0 + sgn * t, its centre staged row c0 + sgn * t.
    const int mid = DUO ? min(ty0 + g.tile_h, g.h) : 0;          // first row of wave 1
    const int sgn = DUO ? (wv ? 1 : -1) : 1;
    const int y0 = DUO ? (wv ? mid : mid - 1) : ty0;
    const int c0 = DUO ? y0 - ty0 + HALF : HALF;
    const int rows_out = DUO ? (wv ? min(ty0 + 2 * g.tile_h, g.h) - mid : mid - ty0)
                             : min(g.tile_h, g.h - ty0);

    // ---- warm-up: the N window rows of output row 0.  A loop of its own, so that the
    // steady-state loop below has ONE code path updating S (with both in one loop the
    // register allocator met two definitions of every sum plane at the join and paid
    // 16 x SB register copies per row for it).
    SM_SLICE_PRIO();                // the warm-up rows are shorter than one time slice
    if (DUO) {
        // this wave's HALF of the N - 1 rows both first windows contain ...
#pragma unroll 1
        for (int e = 0; e <= HALF; e++) {
        if (e == HALF) {
        // ... the other wave's half, through an LDS block [2 halves of the shifts][DS / 2 *
        // SB / 2 plane pairs][64 lanes]: wave w assembles the totals of half w.  Each wave
        // (1) writes its partial sums of the OTHER half, (2) adds the partner's partial sums
        // of its own half to its own and writes the totals back to the same words, (3) reads
        // the totals of the other half -- two barriers.  (A first version swapped two shifts
        // at a time, both ways, with a barrier each: 8 barriers, each of which waits for
        // whichever of the two waves its SIMD currently serves at the lower priority --
        // measured ~2.3 us per workgroup, more than one of the 1.7 us warm-up rows saved.)
        constexpr int DH = DS / 2, XP = DH * SB / 2;
        static_assert(DS % 2 == 0 && (DH * SB) % 2 == 0, "exchange halves");
        typedef u32 v2u __attribute__((ext_vector_type(2)));
        // (the two slots end / begin at word g.xm_off: behind them -- and over them, once the warm-up
        // is done -- lie the buffers of the lane merge, wave w's over the slot that wave w reads LAST)
        v2u *xq = reinterpret_cast<v2u *>(lds + g.xm_off - XP * 128) + tid;
        auto exchange = [&](auto own_tag) {
            constexpr int OWN = decltype(own_tag)::value, OTH = 1 - OWN;
            v2u *slot_own = xq + OWN * XP * 64, *slot_oth = xq + OTH * XP * 64;
#pragma unroll
            for (int j = 0; j < XP; j++) {
                const int p = 2 * j, q = 2 * j + 1;
                const v2u v = {S[OTH * DH + p / SB][p % SB], S[OTH * DH + q / SB][q % SB]};
                slot_oth[j * 64] = v;
            }
            __syncthreads();
#pragma unroll
            for (int dd = 0; dd < DH; dd += 2) {        // two shifts = SB plane pairs at a time
                u32 o[2][SB];
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int p = 2 * j, q = 2 * j + 1;
                    const v2u v = slot_own[(dd * SB / 2 + j) * 64];
                    o[p / SB][p % SB] = v.x;
                    o[q / SB][q % SB] = v.y;
                }
                add_planes<SB, SB>(S[OWN * DH + dd], o[0]);
                add_planes<SB, SB>(S[OWN * DH + dd + 1], o[1]);
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int p = 2 * j, q = 2 * j + 1;
                    const v2u v = {S[OWN * DH + dd + p / SB][p % SB], S[OWN * DH + dd + q / SB][q % SB]};
                    slot_own[(dd * SB / 2 + j) * 64] = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < XP; j++) {
                const int p = 2 * j, q = 2 * j + 1;
                const v2u v = slot_oth[j * 64];
                S[OTH * DH + p / SB][p % SB] = v.x;
                S[OTH * DH + q / SB][q % SB] = v.y;
            }
        };
        if (wv) exchange(IntTag<1>{}); else exchange(IntTag<0>{});
        }
        // ... and (e == HALF) the one row only this wave's first window has
        slide_in(c0 + sgn * e);
        }
        if (rows_out <= 0) return;      // wave 1 of a workgroup that ends at the image's last row
    } else {
#pragma unroll 1
        for (int e = 0; e < N; e++) slide_in(e);
    }
    SM_SLICE_READ();
    SM_STAMP(2);

#if SM_BS_XMERGE
    // lane merge through LDS (see the row loop): this wave's buffer of g.xm_words words = 4 x NPG blocks of 1 KB.  In a two-wave
    // workgroup wave 0's begins at word g.xm_off (over exchange slot 1, which wave 0 is the last to read) and
    // wave 1's ends there (over slot 0); a lone wave's begins there.
    u32 *xm_base = lds + g.xm_off - ((DUO && wv) ? g.xm_words : 0);
    u32 *xm_wbase = xm_base + (s & 3) * ((SB + AB + 3) / 4) * 256;
    const u32 xm_wunit = (u32)(wi * (g.nl >> 2) + (s >> 2) + 2 * (s & 3));
#endif
#if SM_BS_PREFETCH
    // byte addresses (LDS offsets: the low half of the flat address) of this lane's words
    // in staged row 0, and the row strides; the reads of iteration t + 1 are issued while
    // iteration t computes
    const u32 ldsL = (u32)(uintptr_t)pL, ldsR = (u32)(uintptr_t)pR;
    // (signed strides: a DUO workgroup's wave 0 walks up the staged rows)
    const u32 sL = (u32)(sgn * 4 * plw), sR = (u32)(sgn * 4 * prw);
    // the row that slides in next: HALF + 1 beyond the centre row (staged row N of a lone wave)
    u32 aNewL = ldsL + 4u * wL + sL * (HALF + 1) + 4u * plw * c0, aNewR = ldsR + 4u * wR + sR * (HALF + 1) + 4u * prw * c0;
    u32 aCenL = ldsL + 4u * wLc + 4u * plw * c0, aCenR = ldsR + 4u * wRc + 4u * prw * c0;
    RawRow qn, qo;
    RawCentre qc;
    lds_issue(qc, aCenL, aCenR);
    if (!DUO || rows_out > 1) {
        lds_issue(qn, aNewL, aNewR);
        lds_issue(qo, aNewL - sL * N, aNewR - sR * N);
    }
#endif
#pragma unroll 1
    for (int t = 0;;) {
        SM_SLICE_PRIO();
        // ---- winner-take-all of output row t over this lane's 16 shifts
        const int y = y0 + sgn * t;
#if SM_BS_PREFETCH
        const u32 lc = qc.l;
        const u32 c0_ = (u32)qc.r01, c1_ = (u32)(qc.r01 >> 32), c2_ = qc.r2;
        const u32 rc0 = alignbit(c1_, c0_, shRc), rc1 = alignbit(c2_, c1_, shRc);
#else
        const u32 lc = pL[(c0 + sgn * t) * plw + wLc];
        const u32 *rrc = pR + (c0 + sgn * t) * prw + wRc;
        const u32 rc0 = alignbit(rrc[1], rrc[0], shRc), rc1 = alignbit(rrc[2], rrc[1], shRc);
#endif

        u32 B[SB], arg[ABMAX];
#pragma unroll
        for (int k = 0; k < ABMAX; k++) arg[k] = 0;
#if SM_BS_LOCKSTEP
        {
            // Four independent scans side by side, one per quarter of the lane's shifts
            // (ascending, <=: the last of equal counts wins), then the quarters are merged
            // pairwise -- the higher quarter wins ties -- which is the same winner as one
            // scan over all shifts.
            constexpr int Q = 4, QS = DS / Q;          // shifts per quarter
            constexpr int AQ = AB - 2;                 // planes of the index within a quarter
            u32 Bq[Q][SB], aq[Q][AQ > 0 ? AQ : 1];
#pragma unroll
            for (int i = 0; i < QS; i++) {
                u32 upd[Q];
                u32 rcd[Q];
#pragma unroll
                for (int qd = 0; qd < Q; qd++) {
                    const int dd = qd * QS + i;
                    rcd[qd] = dd ? alignbit(rc1, rc0, dd) : rc0; SM_PIN();
                }
                SM_SYNC(1);
                if (i == 0) {
                    // first shift of a quarter: it wins wherever its centre pixel matches
#pragma unroll
                    for (int qd = 0; qd < Q; qd++) {
                        upd[qd] = bop<BOP_XNOR>(lc, rcd[qd], 0u); SM_PIN();
                        if (!FULLD) { upd[qd] &= (u32)__builtin_amdgcn_sbfe((int)dvalid, qd * QS + i, 1); SM_PIN(); }
                    }
#pragma unroll
                    for (int k = 0; k < SB; k++)
#pragma unroll
                        for (int qd = 0; qd < Q; qd++) { Bq[qd][k] = bop<BOP_ORN>(upd[qd], S[qd * QS + i][k], 0u); SM_PIN(); }
#pragma unroll
                    for (int a = 0; a < AQ; a++)
#pragma unroll
                        for (int qd = 0; qd < Q; qd++) aq[qd][a] = 0;
                } else {
                    u32 bw[Q];
#pragma unroll
                    for (int k = 0; k < SB; k++)
#pragma unroll
                        for (int qd = 0; qd < Q; qd++) {
                            bw[qd] = bop<BOP_BORROW>(Bq[qd][k], S[qd * QS + i][k], k ? bw[qd] : 0u); SM_PIN();
                            if (qd == Q - 1 && (k & 1)) SM_SYNC(3);
                        }
#pragma unroll
                    for (int qd = 0; qd < Q; qd++) {
                        upd[qd] = bop<BOP_UPD>(lc, rcd[qd], bw[qd]); SM_PIN();
                        if (!FULLD) { upd[qd] &= (u32)__builtin_amdgcn_sbfe((int)dvalid, qd * QS + i, 1); SM_PIN(); }
                    }
#pragma unroll
                    for (int k = 0; k < SB; k++)
#pragma unroll
                        for (int qd = 0; qd < Q; qd++) {
                            Bq[qd][k] = bop<BOP_SEL>(upd[qd], S[qd * QS + i][k], Bq[qd][k]); SM_PIN();
                            if (qd == Q - 1 && (k & 1)) SM_SYNC(3);
                        }
#pragma unroll
                    for (int a = 0; a < AQ; a++)
#pragma unroll
                        for (int qd = 0; qd < Q; qd++) {
                            if ((i >> a) & 1) aq[qd][a] |= upd[qd]; else aq[qd][a] = bop<BOP_ANDN>(upd[qd], aq[qd][a], 0u);
                            SM_PIN();
                        }
                }
            }
            // quarters 0|1 and 2|3 side by side: take the higher one iff its count is <=
            u32 bwm[2] = {0, 0};
#pragma unroll
            for (int k = 0; k < SB; k++)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    // borrow of lower - higher: 1 <=> lower < higher
                    bwm[m] = bop<BOP_BORROW>(Bq[2 * m][k], Bq[2 * m + 1][k], k ? bwm[m] : 0u); SM_PIN();
                }
            // halves: the selects of one plane feed the final comparison of that plane
            u32 Bh[2][SB], ah[2][AQ + 1];
            u32 bwf = 0;
#pragma unroll
            for (int k = 0; k < SB; k++) {
#pragma unroll
                for (int m = 0; m < 2; m++) { Bh[m][k] = bop<BOP_SEL>(bwm[m], Bq[2 * m][k], Bq[2 * m + 1][k]); SM_PIN(); }
                if (k < AQ) {
#pragma unroll
                    for (int m = 0; m < 2; m++) { ah[m][k] = bop<BOP_SEL>(bwm[m], aq[2 * m][k], aq[2 * m + 1][k]); SM_PIN(); }
                }
                if (k > 0) { bwf = bop<BOP_BORROW>(Bh[0][k - 1], Bh[1][k - 1], k > 1 ? bwf : 0u); SM_PIN(); }
            }
#pragma unroll
            for (int k = SB; k < AQ; k++)
#pragma unroll
                for (int m = 0; m < 2; m++) { ah[m][k] = bop<BOP_SEL>(bwm[m], aq[2 * m][k], aq[2 * m + 1][k]); SM_PIN(); }
            ah[0][AQ] = ~bwm[0]; SM_PIN();             // index bit AQ: the higher quarter was taken
            ah[1][AQ] = ~bwm[1]; SM_PIN();
            bwf = bop<BOP_BORROW>(Bh[0][SB - 1], Bh[1][SB - 1], SB > 1 ? bwf : 0u); SM_PIN();
            // bwf = 1 <=> lower half < higher half: keep the lower one
#pragma unroll
            for (int k = 0; k < SB; k++) { B[k] = bop<BOP_SEL>(bwf, Bh[0][k], Bh[1][k]); SM_PIN(); }
#pragma unroll
            for (int k = 0; k <= AQ; k++) { arg[k] = bop<BOP_SEL>(bwf, ah[0][k], ah[1][k]); SM_PIN(); }
            arg[AQ + 1] = ~bwf; SM_PIN();
        }
#else
#pragma unroll
        for (int k = 0; k < SB; k++) B[k] = 0xffffffffu;
#pragma unroll
        for (int dd = 0; dd < DS; dd++) {
            const u32 rcd = dd ? alignbit(rc1, rc0, dd) : rc0;
            u32 bw = 0;                                  // borrow of B - S: 1 <=> B < S
#pragma unroll
            for (int k = 0; k < SB; k++) bw = bop<BOP_BORROW>(B[k], S[dd][k], bw);
            u32 upd = bop<BOP_UPD>(lc, rcd, bw);         // centre matches and S <= B
            if (!FULLD) upd &= (u32)__builtin_amdgcn_sbfe((int)dvalid, dd, 1);
#pragma unroll
            for (int k = 0; k < SB; k++) B[k] = bop<BOP_SEL>(upd, S[dd][k], B[k]);
#pragma unroll
            for (int k = 0; k < AB; k++) {
                if ((dd >> k) & 1) arg[k] |= upd; else arg[k] &= ~upd;
            }
        }
#endif

        // ---- planes -> integers, for `per` pixels of the word at image column xw, row yy, starting at
        // pixel pfirst of the word, in chunks of up to 4.  nib = the chunk's bits of a plane;
        // nib * 0x204081 puts copy j of the nibble at bit 7j, so bit 8q holds pixel q's bit:
        // & 0x01010101 leaves one byte per pixel, and byte lanes then add up the planes.
        // (pstep: pixels between a lane's consecutive chunks -- 4 where a lane owns a run of `per` pixels; the lanes
        // that share an item of the LDS merge interleave their chunks, so that one store instruction writes
        // 16 * lanes-per-item contiguous bytes of every word)
        auto emit_pixels = [&](u32 (&B)[SB], u32 (&arg)[ABMAX], int xw, int yy, int pfirst, int per, int pstep) {
            const int cw = per < 4 ? per : 4;          // pixels per chunk
            const u32 cmask = (1u << cw) - 1u;
            u32 allone = B[0];
#pragma unroll
            for (int k = 1; k < SB; k++) allone &= B[k];
#if SM_BS_XMERGE
            // "no shift matched" (all planes of the count set) -> web = D: folded in on the PLANES,
            // 32 pixels per operation, instead of a select per pixel (bit k of D - 1 for all lanes)
            u32 dm1 = (u32)(g.D - 1);
            asm volatile("" : "+s"(dm1));       // (re-made here from one SGPR: hoisted out of the row loop the ABMAX
                                                // masks cost SGPR spills, i.e. v_readlane in front of every use)
#pragma unroll
            for (int k = 0; k < ABMAX; k++) {
                if (k >= AB && k >= AB + g.log2nl) continue;   // uniform: planes no merge level set
                arg[k] = bop<BOP_SEL>(allone, (u32)-(i32)((dm1 >> k) & 1u), arg[k]);
            }
#endif
            for (int c = 0, p0 = pfirst; c < per; c += 4, p0 += pstep) {
                u32 bb = 0, bhi = 0, alo = 0, ahi = 0;
                if (best) {                          // uniform: the counts are wanted at all
#pragma unroll
                    for (int k = 0; k < SB; k++) {
                        const u32 sp = __umul24((B[k] >> p0) & cmask, 0x204081u) & 0x01010101u;
                        if (k < 8) bb += sp << k; else bhi += sp << (k - 8);
                    }
                }
#pragma unroll
                for (int k = 0; k < ABMAX; k++) {
                    if (k >= AB && k >= AB + g.log2nl) continue;   // uniform: planes no merge level set
                    const u32 sp = __umul24((arg[k] >> p0) & cmask, 0x204081u) & 0x01010101u;
                    if (k < 8) alo += sp << k; else ahi += sp << (k - 8);
                }
                const u32 none = __umul24((allone >> p0) & cmask, 0x204081u) & 0x01010101u;
                i32 wv[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool no = (none >> (8 * q)) & 1u;
                    int taps = N * N;
                    if (GHOST) {
                        const int x = xw + p0 + q;
                        const int cols = min(g.w - 1, x + HALF) - max(0, x - HALF) + 1;
                        const int rws = min(g.h - 1, yy + HALF) - max(0, yy - HALF) + 1;
                        taps = cols * rws;
                    }
                    const i32 a = (i32)(((alo >> (8 * q)) & 255u) | (((ahi >> (8 * q)) & 255u) << 8));
#if SM_BS_XMERGE
                    wv[q] = a + 1;
#else
                    wv[q] = no ? g.D : a + 1;
#endif
                    bv[q] = no ? 0 : taps - (i32)(((bb >> (8 * q)) & 255u) | (((bhi >> (8 * q)) & 255u) << 8));
                }
                const int x = xw + p0;
                const size_t o = ((size_t)pair * g.h + yy) * g.w + x;
                const bool vec = cw == 4 && g.vec_ok && x + 4 <= g.w;
                // streaming stores: the maps are written once and never read here
                if (g.web_bytes == 4) {
                    if (vec) {
                        const v4i wq = {wv[0], wv[1], wv[2], wv[3]};
                        store_map4(web + o, wq);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (q < cw && x + q < g.w) web[o + q] = wv[q];
                    }
                } else if (g.web_bytes == 1) {
                    // narrow maps (sm_match_wta_typed): the same values as uint8 / uint16
                    u8 *web8 = reinterpret_cast<u8 *>(web);
                    if (vec) {
                        u32 pk = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) pk |= (u32)(wv[q] & 255) << (8 * q);
                        __builtin_nontemporal_store(pk, reinterpret_cast<u32 *>(web8 + o));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (q < cw && x + q < g.w) web8[o + q] = (u8)wv[q];
                    }
                } else {
                    unsigned short *web16 = reinterpret_cast<unsigned short *>(web);
                    if (vec) {
                        typedef u32 v2u __attribute__((ext_vector_type(2)));
                        const v2u pk = {(u32)wv[0] | ((u32)wv[1] << 16), (u32)wv[2] | ((u32)wv[3] << 16)};
                        __builtin_nontemporal_store(pk, reinterpret_cast<v2u *>(web16 + o));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (q < cw && x + q < g.w) web16[o + q] = (unsigned short)wv[q];
                    }
                }
                if (best) {
                    if (vec) {
                        const v4i bq = {bv[0], bv[1], bv[2], bv[3]};
                        store_map4(best + o, bq);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (q < cw && x + q < g.w) best[o + q] = bv[q];
                    }
                }
            }
        };

        // lanes `me` and me ^ (1 << K) hold candidates of the same pixels: lower count wins, on a tie
        // the lane with the higher shifts (bit K of `me` set); the winner's bit K becomes plane NA
#define SM_MERGE(K, me, NA)                                                            \
        {                                                                              \
            u32 pb[SB], pa[NA];                                                        \
            _Pragma("unroll") for (int k = 0; k < SB; k++) pb[k] = from_partner<K>(B[k]);      \
            _Pragma("unroll") for (int k = 0; k < NA; k++) pa[k] = from_partner<K>(arg[k]);    \
            SM_PIN(); SM_SYNC(1);                                                      \
            const u32 mine_high = ((me) >> K) & 1 ? 0xffffffffu : 0u;                  \
            u32 bw = mine_high;          /* borrow-in 1: partner - mine - 1 < 0 <=> partner <= mine */ \
            bw = ~bw;                    /* partner is the high one iff I am not */   \
            _Pragma("unroll") for (int k = 0; k < SB; k++) bw = bop<BOP_BORROW>(pb[k], B[k], bw); \
            const u32 take = bw;                                                       \
            _Pragma("unroll") for (int k = 0; k < SB; k++) B[k] = bop<BOP_SEL>(take, pb[k], B[k]); \
            _Pragma("unroll") for (int k = 0; k < NA; k++) arg[k] = bop<BOP_SEL>(take, pa[k], arg[k]); \
            arg[NA] = take ^ mine_high;   /* partner's bit K = ~mine */              \
        }
#if SM_BS_XMERGE
        if (g.xmerge) {
            // ---- THE NL LANES OF A WORD ARE MERGED THROUGH LDS, FOUR ROWS AT A TIME.  Merging them per
            // row with DPP costs log2(nl) full compare-and-select levels per lane and row (~105 VALU
            // instructions of 1640 at C3, 36 of them half-rate DPP moves; measured by leaving them out:
            // -7 % of the launch).  Instead every lane hands the planes of its candidate over (NPG
            // 16-byte stores per row), and after four rows the 4 * 64 / nl (row, word) items of the batch
            // are dealt to the 64 lanes, nl / 4 lanes per item: each scans FOUR candidates in ascending
            // shift order (<=: the later of equal counts wins) and only log2(nl) - 2 DPP levels remain,
            // once per four rows; lane `sub` of an item then turns 128 / nl pixels into integers.
            // Layout: block (j, pg) of 64 x 16 bytes holds plane group pg of the candidates of the
            // shift lanes s = 4 * sub + j; candidate (row r, word wi, sub) sits at unit
            // (16 r + wi * LPI + sub + 2 j) & 63 -- a reader lane l reads unit (l + 2 j) & 63 of each
            // block (one aligned 1 KB run per read: conflict-free), a writer's 8 contiguous lanes hit 8
            // distinct 16-byte columns of the 32 store banks.
            constexpr int NPL = SB + AB, NPG = (NPL + 3) / 4;
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) const volatile v4u lds_vv4;
            const u32 xm_lds = (u32)(uintptr_t)xm_base;          // LDS offset: the low half of the flat address
            const int r = t & 3;
            {
                u32 pl[4 * NPG];
#pragma unroll
                for (int k = 0; k < 4 * NPG; k++) pl[k] = k < SB ? B[k] : (k < NPL ? arg[k - SB] : 0u);
                const u32 unit = (xm_wunit + 16u * (u32)r) & 63u;
#pragma unroll
                for (int pg = 0; pg < NPG; pg++) {
                    const v4u v = {pl[4 * pg], pl[4 * pg + 1], pl[4 * pg + 2], pl[4 * pg + 3]};
                    *reinterpret_cast<v4u *>(xm_wbase + pg * 256 + unit * 4) = v;
                }
            }
            if (r == 3 || t + 1 >= rows_out) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // reader role: item (rr, wr), sub-lane `sub` of its LPI lanes
                const int lpi = g.nl >> 2, l2lpi = g.log2nl - 2;
                const int rr = tid >> 4, wr = (tid & 15) >> l2lpi, sub = tid & (lpi - 1);
                u32 Bm[SB], am[ABMAX];
#pragma unroll
                for (int k = 0; k < ABMAX; k++) am[k] = 0;
                // all 4 x NPG reads are issued before the first candidate is looked at (volatile: they stay
                // where they are written, as the row prefetch above): ONE exposed LDS latency per batch --
                // left to itself the compiler reads candidate by candidate and the wave waits seven times
                v4u ev[4][NPG];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 unit = ((u32)tid + 2u * j) & 63u;
#pragma unroll
                    for (int pg = 0; pg < NPG; pg++)
                        ev[j][pg] = *(lds_vv4 *)(uintptr_t)(xm_lds + 4u * (u32)((j * NPG + pg) * 256) + 16u * unit);
                }
                SM_PIN();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    u32 e[4 * NPG];
#pragma unroll
                    for (int pg = 0; pg < NPG; pg++) {
                        const v4u v = ev[j][pg];
                        e[4 * pg] = v.x; e[4 * pg + 1] = v.y; e[4 * pg + 2] = v.z; e[4 * pg + 3] = v.w;
                    }
                    if (j == 0) {
#pragma unroll
                        for (int k = 0; k < SB; k++) Bm[k] = e[k];
#pragma unroll
                        for (int k = 0; k < AB; k++) am[k] = e[SB + k];
                    } else {
                        u32 bw = 0xffffffffu;      // borrow-in 1: candidate - mine - 1 < 0 <=> candidate <= mine
#pragma unroll
                        for (int k = 0; k < SB; k++) bw = bop<BOP_BORROW>(e[k], Bm[k], bw);
                        const u32 take = bw;
#pragma unroll
                        for (int k = 0; k < SB; k++) Bm[k] = bop<BOP_SEL>(take, e[k], Bm[k]);
#pragma unroll
                        for (int k = 0; k < AB; k++) am[k] = bop<BOP_SEL>(take, e[SB + k], am[k]);
                        if (j & 1) am[AB] |= take; else am[AB] = bop<BOP_ANDN>(take, am[AB], 0u);
                        if (j & 2) am[AB + 1] |= take; else am[AB + 1] = bop<BOP_ANDN>(take, am[AB + 1], 0u);
                    }
                }
                {
                    u32 (&B)[SB] = Bm;
                    u32 (&arg)[ABMAX] = am;
                    if (lpi > 1) SM_MERGE(0, sub, AB + 2)
                    if (lpi > 2) SM_MERGE(1, sub, AB + 3)
                    if (lpi > 4) SM_MERGE(2, sub, AB + 4)
                }
                const int tt = t - r + rr;                      // the item's output row of this wave
                const int yy = y0 + sgn * tt;
                // (The lanes of an item interleave their 4-pixel chunks: one store instruction writes 16 * lpi
                // contiguous bytes of every word -- with a RUN of pixels per lane the 16-byte pieces lay 64 bytes
                // apart and the write-through stores moved every 32-byte sector twice: WRITE_SIZE 64 800 KiB per C3
                // launch for a 32 400 KiB map.  Routing the integers through LDS so that every store writes whole
                // rows was built and measured -- no better at C3 / C5 (90.6 vs 89.5 us), the extra LDS round trip
                // costs what the contiguous stores gain; the store flavour matters far more: with `nt` instead of
                // `sc1` scattered pieces cost a factor 2.3 at 8 x 1080p.  profiles/r04/ab_store_flavour.txt)
                if (rr <= r && yy < g.h && yy >= 0) {
                    const int per = 32 >> l2lpi;
                    emit_pixels(Bm, am, tx0 + 32 * wr, yy, 4 * sub, per, 4 * lpi);
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else
#endif
        {
            // ---- merge the nl lanes of this word per row (DPP; ds_bpermute beyond 16 lanes)
            if (g.nl > 1) SM_MERGE(0, s, AB)
            if (g.nl > 2) SM_MERGE(1, s, AB + 1)
            if (g.nl > 4) SM_MERGE(2, s, AB + 2)
            if (g.nl > 8) SM_MERGE(3, s, AB + 3)
            if (g.nl > 16) SM_MERGE(4, s, AB + 4)
            if (g.nl > 32) SM_MERGE(5, s, AB + 5)
            // After the merge all nl lanes of a word hold the same planes; lane s converts
            // pixels [s * per, s * per + per), per = 32 / nl
            if (y < g.h) {
                const int per = 32 >> g.log2nl;            // nl <= 32
                emit_pixels(B, arg, x0, y, s * per, per, 4);
            }
        }
#undef SM_MERGE

        // ---- slide the window down: staged row t + N - 1 in, staged row t - 1 out
        SM_SYNC(1);
        if (++t >= rows_out) break;
#if SM_BS_PREFETCH
        {
            // cut the views of this slide out of the raw words, which frees their
            // registers for the reads of the NEXT iteration: a whole slide ahead of use
            RowViews vn, vo;
            views_of(qn, vn);
            views_of(qo, vo);
            aNewL += sL; aNewR += sR; aCenL += sL; aCenR += sR;
            lds_issue(qc, aCenL, aCenR);
            if (t + 1 < rows_out) {               // uniform; the last row slides no further
                lds_issue(qn, aNewL, aNewR);
                lds_issue(qo, aNewL - sL * N, aNewR - sR * N);
            }
            SM_SLICE_READ();         // behind the reads above: it is waited for with them, a row later
            SM_PIN(); SM_SYNC(1);
            slide_views(vn, vo);
        }
#else
        slide_both(c0 + sgn * (t + HALF), c0 + sgn * (t - 1 - HALF));
#endif
    }
    SM_STAMP(3);
}


template <int N, int DS, bool CAP2, bool DUO = false>
static const void *bs_ptr4(bool fulld, bool ghost)
{
    return fulld ? (ghost ? (const void *)k_match_bs<N, DS, true, true, CAP2, DUO> : (const void *)k_match_bs<N, DS, true, false, CAP2, DUO>)
                 : (ghost ? (const void *)k_match_bs<N, DS, false, true, CAP2, DUO> : (const void *)k_match_bs<N, DS, false, false, CAP2, DUO>);
}
// CAPPABLE: this window's kernel needs few enough registers for three waves per SIMD,
// so a two-wave variant is built next to it
template <int N, int DS, bool CAPPABLE>
static const void *bs_ptr(bool fulld, bool ghost, bool cap2)
{
    if (cap2) return CAPPABLE ? bs_ptr4<N, DS, CAPPABLE>(fulld, ghost) : nullptr;
    return bs_ptr4<N, DS, false>(fulld, ghost);
}
