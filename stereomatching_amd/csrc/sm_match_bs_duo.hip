// sm_match_bs_duo.hip -- builds of the bit-sliced kernel with two-wave workgroups (shared
// warm-up, see sm_match_bs_kernel.h), 16 shifts per lane.

#define SM_BS_TU duo
#include "sm_match_bs_kernel.h"

const void *sm_bs_ptr_duo(int n, bool fulld, bool ghost)
{
    switch (n) {
    case 3: return bs_ptr4<3, 16, true, true>(fulld, ghost);
    case 5: return bs_ptr4<5, 16, true, true>(fulld, ghost);
    case 7: return bs_ptr4<7, 16, true, true>(fulld, ghost);
    case 9: return bs_ptr4<9, 16, false, true>(fulld, ghost);
    case 11: return bs_ptr4<11, 16, false, true>(fulld, ghost);
    default: return nullptr;
    }
}
