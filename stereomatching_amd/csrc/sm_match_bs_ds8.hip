// sm_match_bs_ds8.hip -- builds of the bit-sliced kernel with 8 shifts per lane, one wave per
// workgroup (a translation unit of its own so that the library's parts compile side by side).

#define SM_BS_TU ds8
#include "sm_match_bs_kernel.h"

const void *sm_bs_ptr_ds8(int n, bool fulld, bool ghost, bool cap2)
{
    switch (n) {
    case 3: return bs_ptr<3, 8, true>(fulld, ghost, cap2);
    case 5: return bs_ptr<5, 8, true>(fulld, ghost, cap2);
    case 7: return bs_ptr<7, 8, true>(fulld, ghost, cap2);
    case 9: return bs_ptr<9, 8, true>(fulld, ghost, cap2);
    case 11: return bs_ptr<11, 8, true>(fulld, ghost, cap2);
    case 13: return bs_ptr<13, 8, true>(fulld, ghost, cap2);
    case 15: return bs_ptr<15, 8, true>(fulld, ghost, cap2);
    case 17: return bs_ptr<17, 8, false>(fulld, ghost, cap2);
    case 19: return bs_ptr<19, 8, false>(fulld, ghost, cap2);
    case 21: return bs_ptr<21, 8, false>(fulld, ghost, cap2);
    default: return nullptr;
    }
}
