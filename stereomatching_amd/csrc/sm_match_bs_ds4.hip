// sm_match_bs_ds4.hip -- builds of the bit-sliced kernel with 4 shifts per lane (for grids that leave the
// chip mostly empty -- a single small pair -- and for tall windows with few shifts, whose warm-up rows
// weigh less on narrower, taller tiles), one-wave and two-wave workgroups.

#define SM_BS_TU ds4
#include "sm_match_bs_kernel.h"

const void *sm_bs_ptr_ds4(int n, bool fulld, bool ghost, bool cap2, bool duo)
{
    if (duo) {
        if (cap2) return nullptr;
        switch (n) {
        case 3: return bs_ptr4<3, 4, true, true>(fulld, ghost);
        case 5: return bs_ptr4<5, 4, true, true>(fulld, ghost);
        case 7: return bs_ptr4<7, 4, true, true>(fulld, ghost);
        case 9: return bs_ptr4<9, 4, true, true>(fulld, ghost);
        case 11: return bs_ptr4<11, 4, true, true>(fulld, ghost);
        case 13: return bs_ptr4<13, 4, true, true>(fulld, ghost);
        case 15: return bs_ptr4<15, 4, true, true>(fulld, ghost);
        case 17: return bs_ptr4<17, 4, true, true>(fulld, ghost);
        case 19: return bs_ptr4<19, 4, true, true>(fulld, ghost);
        case 21: return bs_ptr4<21, 4, true, true>(fulld, ghost);
        default: return nullptr;
        }
    }
    switch (n) {
    case 3: return bs_ptr<3, 4, true>(fulld, ghost, cap2);
    case 5: return bs_ptr<5, 4, true>(fulld, ghost, cap2);
    case 7: return bs_ptr<7, 4, true>(fulld, ghost, cap2);
    case 9: return bs_ptr<9, 4, true>(fulld, ghost, cap2);
    case 11: return bs_ptr<11, 4, true>(fulld, ghost, cap2);
    case 13: return bs_ptr<13, 4, true>(fulld, ghost, cap2);
    case 15: return bs_ptr<15, 4, true>(fulld, ghost, cap2);
    case 17: return bs_ptr<17, 4, true>(fulld, ghost, cap2);
    case 19: return bs_ptr<19, 4, true>(fulld, ghost, cap2);
    case 21: return bs_ptr<21, 4, true>(fulld, ghost, cap2);
    default: return nullptr;
    }
}
