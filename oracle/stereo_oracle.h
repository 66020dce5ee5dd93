/*
 * stereo_oracle.h -- CPU restatement of the reference's stereo pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and there only as the checker (never as the thing measured or
 * shipped).  The product path is stereomatching_amd/csrc (HIP) behind
 * include/stereo_hip.h.
 *
 * What it restates (all paths relative to /root/reference):
 *   src/stereo.c:16-84        find_all_edges (toroidal borders)
 *   src/stereo-ghost.c:18-85  find_all_edges (ghost borders, 128.0 halo)
 *   src/stereo.c:113-127      fillup_matches
 *   src/stereo.c:132-148      addup_pixels_in_square
 *   src/stereo.c:172-192      record_score / fillup_scores
 *   src/stereo.c:196-220      find_highest_scoring_shifts
 *   src/stereo.c:230-274      fill_web_holes / draw_contour_map
 * generalised over D (the reference's compile-time NUM_SHIFTS = 30,
 * src/stereo.c:6) and S (square_width, src/stereo.c:365).
 *
 * Parity status: PINNED.  oracle/Makefile builds the unmodified reference
 * sources into oracle/_ref/ and tests/test_oracle_vs_ref.py checks every
 * stage of this file bit-for-bit against it at D = 30; the committed
 * fixtures under tests/golden/ were produced by that compiled reference
 * (tests/golden/make_golden.py).
 */
#ifndef STEREO_ORACLE_H
#define STEREO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SMO_TOROIDAL = 0, SMO_GHOST = 1 };

/* step 1: u8 gray (brightness = k/256.0, src/image.c:9-15) -> u8 edges {0,1} */
void smo_find_all_edges(const uint8_t *gray, int w, int h, double threshold,
                        int mode, uint8_t *edges);

/* one (Sl,Sr) decision of the 3-vs-3 contrast test; sums are in units of
 * 1/256 (the ghost halo 128.0 counts as 32768).  Used to pin the GPU's
 * integer-keyed edge test exhaustively. */
int smo_edge_decision(int sum_left, int sum_right, double threshold);

/* table[sa*766+sb] = smo_edge_decision(sa, sb, threshold) for all in-image sums */
void smo_edge_table(double threshold, uint8_t *table);

/* step 2, plane at a time (what the reference keeps in matches[i]/scores[i]) */
void smo_match_plane(const uint8_t *left_edges, const uint8_t *right_edges,
                     int w, int h, int shift, int mode, uint8_t *match);
/* faithful loop nest (window tap outermost, modulo indexing) */
void smo_addup_faithful(const uint8_t *match, int w, int h, int square_width,
                        int mode, int32_t *total);
/* same result by separable running sums; for tests at sizes the faithful
 * loop cannot finish in seconds */
void smo_addup_fast(const uint8_t *match, int w, int h, int square_width,
                    int mode, int32_t *total);
void smo_record_score(const uint8_t *match, const int32_t *sum, int w, int h,
                      int32_t *score /* pre-zeroed */);

/* step 2 end to end: best score and winning shift (1..D) per pixel.
 * faithful != 0 selects smo_addup_faithful.  Streams one plane at a time
 * (O(w*h) memory) but the arithmetic is the reference's. */
void smo_hot_path(const uint8_t *left_edges, const uint8_t *right_edges,
                  int w, int h, int num_shifts, int square_width, int mode,
                  int faithful, int32_t *best, int32_t *web);

/* ---- SAD / SSD cost mode: NOT in the reference ("parity unpinned") --------
 * BASELINE.json words the hot path as "SAD/SSD per-pixel cost ... window
 * aggregation ... winner-take-all argmin"; the reference implements the
 * edge-equality cost above and nothing else.  This is the
 * build's own definition of that mode, on the same skeleton, and the only
 * thing the GPU's SAD mode is checked against:
 *   c_d(x,y) = |L(x,y) - R(x+d,y)|          (SAD)   or its square (SSD), on the
 *              uint8 gray images; R wraps (toroidal) or reads 0 past the right
 *              border (ghost), like the edge images of the reference
 *   A_d      = sum of c_d over the n x n window (wrapping, or taps outside the
 *              image counting 0)
 *   best     = min_d A_d ;  web = 1 + min{ d : A_d == best }   (first shift wins)
 */
enum { SMO_COST_SAD = 1, SMO_COST_SSD = 2 };
void smo_cost_hot_path(const uint8_t *left, const uint8_t *right, int w, int h,
                       int num_shifts, int square_width, int mode, int cost,
                       int32_t *best, int32_t *web);

/* step 3 */
void smo_fill_web_holes(int32_t *web, int w, int h, int times);
/* returns 0, or -1 when the reference would divide by zero */
int smo_draw_contour_map(const int32_t *web, int w, int h, int num_lines,
                         uint8_t *out);

/* whole pipeline on u8 inputs; any output pointer may be NULL */
int smo_pipeline(const uint8_t *left, const uint8_t *right, int w, int h,
                 double threshold, int num_shifts, int square_width, int times,
                 int lines, int mode, int faithful, uint8_t *edges_l,
                 uint8_t *edges_r, int32_t *best, int32_t *web1,
                 int32_t *web2, uint8_t *out);

/* wall-clock seconds, CLOCK_MONOTONIC like src/util.h:102-107 */
double smo_time(void);

#ifdef __cplusplus
}
#endif
#endif
