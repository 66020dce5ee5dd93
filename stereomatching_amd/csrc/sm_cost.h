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
    int lds_bytes;
};

// sm_cost_qs.hip: fills *g and returns the kernel for this plan, or nullptr if the shape is not built
const void *sm_sad_qs_configure(const sm_plan *plan, int pairs, const void *d_left, const void *d_right, SadGeom *g);
