// sm_match_bs_duo8.hip -- builds of the bit-sliced kernel with two-wave workgroups (shared
// warm-up, see sm_match_bs_kernel.h), 8 shifts per lane.

#define SM_BS_TU duo8
#include "sm_match_bs_kernel.h"

const void *sm_bs_ptr_duo8(int n, bool fulld, bool ghost)
{
    switch (n) {
    case 3: return bs_ptr4<3, 8, true, true>(fulld, ghost);
    case 5: return bs_ptr4<5, 8, true, true>(fulld, ghost);
    case 7: return bs_ptr4<7, 8, true, true>(fulld, ghost);
    case 9: return bs_ptr4<9, 8, true, true>(fulld, ghost);
    case 11: return bs_ptr4<11, 8, true, true>(fulld, ghost);
    case 13: return bs_ptr4<13, 8, true, true>(fulld, ghost);
    case 15: return bs_ptr4<15, 8, true, true>(fulld, ghost);
    case 17: return bs_ptr4<17, 8, false, true>(fulld, ghost);
    case 19: return bs_ptr4<19, 8, false, true>(fulld, ghost);
    case 21: return bs_ptr4<21, 8, false, true>(fulld, ghost);
    default: return nullptr;
    }
}
