// sm_match_bs.hip -- the bit-sliced hot-path kernel (sm_match_bs_kernel.h): its builds with
// one wave per workgroup and 16 shifts per lane, and the host side of all of them.

#define SM_BS_TU main
#include "sm_match_bs_kernel.h"

// the other translation units' builds
const void *sm_bs_ptr_ds8(int n, bool fulld, bool ghost, bool cap2);
const void *sm_bs_ptr_duo(int n, bool fulld, bool ghost);
const void *sm_bs_ptr_duo8(int n, bool fulld, bool ghost);
const void *sm_bs_ptr_ds4(int n, bool fulld, bool ghost, bool cap2, bool duo);
#ifdef SM_STAMPS
int sm_bs_set_stamps_ds8(void *buf);
int sm_bs_set_stamps_duo(void *buf);
int sm_bs_set_stamps_duo8(void *buf);
int sm_bs_set_stamps_ds4(void *buf);
extern "C" int sm_debug_set_stamps(void *buf)
{
    int rc = sm_bs_set_stamps_main(buf);
    if (!rc) rc = sm_bs_set_stamps_ds8(buf);
    if (!rc) rc = sm_bs_set_stamps_duo(buf);
    if (!rc) rc = sm_bs_set_stamps_duo8(buf);
    if (!rc) rc = sm_bs_set_stamps_ds4(buf);
    return rc;
}
#endif

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Built combinations.  16 shifts per lane up to 11 x 11 (the 16 x SB sum planes fit
// two waves per SIMD); 8 per lane for the larger windows (9 planes per sum) and, for the
// smaller ones, as the alternative for grids that would leave SIMDs with a single wave.
// cap2: the two-waves-per-SIMD variant (nullptr where the kernel is limited to two waves
// by its registers anyway).  duo: two-wave workgroups -- one build per window, limited to
// two waves per SIMD wherever the registers would allow three (cap2 is not a choice there).
const void *sm_bs_kernel_ptr(int n, int ds, bool fulld, bool ghost, bool cap2, bool duo)
{
    if (ds == 4) return sm_bs_ptr_ds4(n, fulld, ghost, cap2, duo);
    if (duo) {
        if (cap2) return nullptr;
        return ds == 16 ? sm_bs_ptr_duo(n, fulld, ghost) : ds == 8 ? sm_bs_ptr_duo8(n, fulld, ghost) : nullptr;
    }
    if (ds == 8) return sm_bs_ptr_ds8(n, fulld, ghost, cap2);
    if (ds == 16) {
        switch (n) {
        case 3: return bs_ptr<3, 16, true>(fulld, ghost, cap2);
        case 5: return bs_ptr<5, 16, true>(fulld, ghost, cap2);
        case 7: return bs_ptr<7, 16, true>(fulld, ghost, cap2);
        case 9: return bs_ptr<9, 16, false>(fulld, ghost, cap2);
        case 11: return bs_ptr<11, 16, false>(fulld, ghost, cap2);
        default: return nullptr;
        }
    }
    return nullptr;
}

// shifts per lane the plan should use for this window (0: not built)
int sm_bs_default_ds(int n)
{
    if (sm_bs_kernel_ptr(n, 16, true, false, false)) return 16;
    if (sm_bs_kernel_ptr(n, 8, true, false, false)) return 8;
    return 0;
}

// the time-sliced priority schedule of the kernel (SM_BS_PATTERN in sm_match_bs_kernel.h)
unsigned sm_bs_default_pattern(bool duo)
{
    (void)duo;
    return SM_BS_PATTERN;
}

// One launch of the plan's kernel that does nothing (web == nullptr): the runtime loads a
// code object when a kernel of it is first LAUNCHED -- its code object is ~1-2 MB and
// the first real launch otherwise waits ~130 us for it inside the caller's timed region.
int sm_bs_prepare(sm_plan *plan)
{
    const MatchGeom &g = plan->g;
    const void *fn = sm_bs_kernel_ptr(g.n, g.ds, g.nl * g.ds == g.D, plan->border == SM_GHOST, g.cap2 != 0, g.duo != 0);
    if (!fn) return SM_OK;
    i32 *none = nullptr;
    void *args[] = {(void *)&plan->d_ext, (void *)&none, (void *)&none, (void *)&g};
    hipError_t e = hipLaunchKernel(fn, dim3(1, 1, 1), dim3(g.threads), args, g.lds_bytes, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
        return sm_fail(SM_ERR_HIP, "set-up launch of k_match_bs failed: %s", hipGetErrorString(e));
    return SM_OK;
}

int sm_bs_launch(const sm_plan *plan, const MatchLaunch &l, int pairs, i32 *d_web, i32 *d_best, hipStream_t st)
{
    const MatchGeom &g = l.g;
    const void *fn = sm_bs_kernel_ptr(g.n, g.ds, g.nl * g.ds == g.D, plan->border == SM_GHOST, g.cap2 != 0, g.duo != 0);
    if (!fn) return sm_fail(SM_ERR_ARG, "bit-sliced kernel not built for n = %d, %d shifts/lane", g.n, g.ds);
    void *args[] = {(void *)&plan->d_ext, (void *)&d_web, (void *)&d_best, (void *)&g};
    hipError_t e;
    if (l.ev_begin)                 // a timed launch: the events ride on the dispatch (sm_match_wta_typed)
        e = hipExtLaunchKernel(fn, dim3(g.tiles_x, g.tiles_y, pairs), dim3(g.threads), args, g.lds_bytes, st,
                               l.ev_begin, l.ev_end, 0);
    else
        e = hipLaunchKernel(fn, dim3(g.tiles_x, g.tiles_y, pairs), dim3(g.threads), args, g.lds_bytes, st);
    if (e != hipSuccess)
        return sm_fail(SM_ERR_HIP, "launch of k_match_bs failed: %s", hipGetErrorString(e));
    return SM_OK;
}
