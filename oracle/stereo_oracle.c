/*
 * stereo_oracle.c -- CPU restatement of the reference pipeline (see
 * stereo_oracle.h for the file:line map and the "test infrastructure only"
 * rule).  Plain C11, no dependencies beyond libc/libm.
 */
#define _POSIX_C_SOURCE 200809L
#include "stereo_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef uint8_t u8;
typedef int32_t i32;

static void *zalloc(size_t n)
{
    void *p = calloc(n ? n : 1, 1);
    if (!p)
        abort();
    return p;
}

/* src/util.h:42-47 -- wrap one coordinate; valid for v >= -m */
static inline int wrap(int v, int m) { return (v + m) % m; }

double smo_time(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + (double)ts.tv_nsec / 1e9;
}

/* ------------------------------------------------------------------ */
/* step 1: edges                                                       */
/* ------------------------------------------------------------------ */

/* the shared tail of the four orientation tests, src/stereo.c:19-27:
 * three-neighbour means on each side, contrast against a threshold that is
 * proportional to the overall brightness and clamped to [0,1] */
static inline int contrast_test(double side_a, double side_b, double threshold)
{
    double mean_a = side_a / 3.0;
    double mean_b = side_b / 3.0;
    double overall = (mean_a + mean_b) / 2.0;
    double limit = threshold * overall;
    if (!(limit > 0.0))
        limit = 0.0;
    if (!(limit < 1.0))
        limit = 1.0;
    return fabs(mean_a - mean_b) > limit;
}

int smo_edge_decision(int sum_left, int sum_right, double threshold)
{
    /* every brightness is k/256 (or the ghost halo 128.0 = 32768/256), so a
     * three-term sum is exact and equals (k1+k2+k3)/256 */
    return contrast_test((double)sum_left / 256.0, (double)sum_right / 256.0,
                         threshold);
}

void smo_edge_table(double threshold, uint8_t *table)
{
    for (int sa = 0; sa < 766; sa++)
        for (int sb = 0; sb < 766; sb++)
            table[sa * 766 + sb] = (u8)smo_edge_decision(sa, sb, threshold);
}

/* neighbour offsets {dx,dy} of the two sides of each orientation:
 * src/stereo.c:19-24 (left|right), :33-38 (top|bottom),
 * :47-52 (up-left|down-right), :61-66 (down-left|up-right) */
static const int8_t ORIENT[4][2][3][2] = {
    {{{-1, -1}, {-1, 0}, {-1, 1}}, {{1, -1}, {1, 0}, {1, 1}}},
    {{{-1, -1}, {0, -1}, {1, -1}}, {{-1, 1}, {0, 1}, {1, 1}}},
    {{{-1, -1}, {0, -1}, {-1, 0}}, {{1, 0}, {0, 1}, {1, 1}}},
    {{{-1, 1}, {0, 1}, {-1, 0}}, {{0, -1}, {1, -1}, {1, 0}}},
};

void smo_find_all_edges(const uint8_t *gray, int w, int h, double threshold,
                        int mode, uint8_t *edges)
{
    /* brightness image with a one-pixel frame: wrapped copies (toroidal,
     * idx() of src/util.h:42) or 128.0 (src/stereo-ghost.c:384-385) */
    const int pw = w + 2;
    double *b = zalloc(sizeof(double) * (size_t)pw * (size_t)(h + 2));
    for (int y = -1; y <= h; y++) {
        for (int x = -1; x <= w; x++) {
            double v;
            int inside = x >= 0 && x < w && y >= 0 && y < h;
            if (inside || mode == SMO_TOROIDAL)
                v = gray[(size_t)wrap(y, h) * w + wrap(x, w)] / 256.0;
            else
                v = 128.0;
            b[(size_t)(y + 1) * pw + (x + 1)] = v;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int edge = 0;
            for (int o = 0; o < 4 && !edge; o++) {
                double side[2];
                for (int s = 0; s < 2; s++) {
                    const int8_t(*n)[2] = ORIENT[o][s];
                    side[s] = b[(size_t)(y + 1 + n[0][1]) * pw + (x + 1 + n[0][0])]
                            + b[(size_t)(y + 1 + n[1][1]) * pw + (x + 1 + n[1][0])]
                            + b[(size_t)(y + 1 + n[2][1]) * pw + (x + 1 + n[2][0])];
                }
                edge = contrast_test(side[0], side[1], threshold);
            }
            edges[(size_t)y * w + x] = (u8)edge;
        }
    }
    free(b);
}

/* ------------------------------------------------------------------ */
/* step 2: match -> window sum -> masked score -> winner               */
/* ------------------------------------------------------------------ */

void smo_match_plane(const uint8_t *le, const uint8_t *re, int w, int h,
                     int shift, int mode, uint8_t *match)
{
    /* src/stereo.c:113-127; ghost: the right edge image is framed with
     * zeros (src/stereo-ghost.c:286-287), so a shifted read past the right
     * border sees 0 */
    for (int y = 0; y < h; y++) {
        const u8 *lrow = le + (size_t)y * w;
        const u8 *rrow = re + (size_t)y * w;
        u8 *mrow = match + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            u8 r;
            if (mode == SMO_TOROIDAL)
                r = rrow[wrap(x + shift, w)];
            else
                r = (x + shift < w) ? rrow[x + shift] : 0;
            mrow[x] = (u8)(lrow[x] == r);
        }
    }
}

void smo_addup_faithful(const uint8_t *match, int w, int h, int square_width,
                        int mode, int32_t *total)
{
    /* src/stereo.c:132-148: the window tap is the OUTER loop and every tap
     * sweeps the whole image, adding into total[] (caller pre-zeroes) */
    const int half = square_width / 2;
    for (int ty = -half; ty <= half; ty++) {
        for (int tx = -half; tx <= half; tx++) {
            for (int y = 0; y < h; y++) {
                for (int x = 0; x < w; x++) {
                    int v;
                    if (mode == SMO_TOROIDAL) {
                        v = match[(size_t)wrap(y + ty, h) * w + wrap(x + tx, w)];
                    } else {
                        int xx = x + tx, yy = y + ty;
                        v = (xx >= 0 && xx < w && yy >= 0 && yy < h)
                                ? match[(size_t)yy * w + xx] : 0;
                    }
                    total[(size_t)y * w + x] += v;
                }
            }
        }
    }
}

void smo_addup_fast(const uint8_t *match, int w, int h, int square_width,
                    int mode, int32_t *total)
{
    /* same sums by a horizontal pass into tmp and a vertical pass out of it */
    const int half = square_width / 2;
    i32 *tmp = zalloc(sizeof(i32) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const u8 *row = match + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            i32 s = 0;
            for (int t = -half; t <= half; t++) {
                int xx = x + t;
                if (mode == SMO_TOROIDAL)
                    s += row[wrap(xx, w)];
                else if (xx >= 0 && xx < w)
                    s += row[xx];
            }
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            i32 s = 0;
            for (int t = -half; t <= half; t++) {
                int yy = y + t;
                if (mode == SMO_TOROIDAL)
                    s += tmp[(size_t)wrap(yy, h) * w + x];
                else if (yy >= 0 && yy < h)
                    s += tmp[(size_t)yy * w + x];
            }
            total[(size_t)y * w + x] += s;
        }
    }
    free(tmp);
}

void smo_record_score(const uint8_t *match, const int32_t *sum, int w, int h,
                      int32_t *score)
{
    /* src/stereo.c:172-182: a score exists only where the pixel itself matched */
    for (size_t p = 0; p < (size_t)w * h; p++)
        if (match[p] == 1)
            score[p] = sum[p];
}

void smo_hot_path(const uint8_t *le, const uint8_t *re, int w, int h,
                  int num_shifts, int square_width, int mode, int faithful,
                  int32_t *best, int32_t *web)
{
    const size_t n = (size_t)w * h;
    u8 *match = zalloc(n);
    i32 *sum = zalloc(sizeof(i32) * n);
    i32 *score = zalloc(sizeof(i32) * n);
    i32 *best_l = best ? best : zalloc(sizeof(i32) * n);

    /* The reference makes two sweeps over the stored score planes
     * (src/stereo.c:201-219): running maximum from 0, then the LAST shift
     * whose score equals the maximum wins, recorded as shift+1.  One sweep
     * with ">=" gives the same pair without keeping the planes: whenever a
     * later plane reaches the running maximum it becomes the winner, and the
     * final winner is therefore the last plane equal to the final maximum. */
    memset(best_l, 0, sizeof(i32) * n);
    for (size_t p = 0; p < n; p++)
        web[p] = 0;
    for (int d = 0; d < num_shifts; d++) {
        smo_match_plane(le, re, w, h, d, mode, match);
        memset(sum, 0, sizeof(i32) * n);
        if (faithful)
            smo_addup_faithful(match, w, h, square_width, mode, sum);
        else
            smo_addup_fast(match, w, h, square_width, mode, sum);
        memset(score, 0, sizeof(i32) * n);
        smo_record_score(match, sum, w, h, score);
        for (size_t p = 0; p < n; p++) {
            if (score[p] >= best_l[p]) {
                best_l[p] = score[p];
                web[p] = d + 1;
            }
        }
    }
    free(match);
    free(sum);
    free(score);
    if (!best)
        free(best_l);
}

/* ------------------------------------------------------------------ */
/* SAD / SSD cost mode (no reference implementation: parity unpinned)  */
/* ------------------------------------------------------------------ */

void smo_cost_hot_path(const uint8_t *left, const uint8_t *right, int w, int h,
                       int num_shifts, int square_width, int mode, int cost,
                       int32_t *best, int32_t *web)
{
    const size_t n = (size_t)w * h;
    const int half = square_width / 2;
    i32 *c = zalloc(sizeof(i32) * n), *t = zalloc(sizeof(i32) * n);
    for (size_t p = 0; p < n; p++) { best[p] = INT_MAX; web[p] = 0; }
    for (int d = 0; d < num_shifts; d++) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int r;
                if (mode == SMO_TOROIDAL) r = right[(size_t)y * w + wrap(x + d, w)];
                else r = x + d < w ? right[(size_t)y * w + x + d] : 0;
                int diff = (int)left[(size_t)y * w + x] - r;
                c[(size_t)y * w + x] = cost == SMO_COST_SSD ? diff * diff : abs(diff);
            }
        for (int y = 0; y < h; y++)          /* horizontal pass */
            for (int x = 0; x < w; x++) {
                i32 s = 0;
                for (int k = -half; k <= half; k++) {
                    int xx = x + k;
                    if (mode == SMO_TOROIDAL) s += c[(size_t)y * w + wrap(xx, w)];
                    else if (xx >= 0 && xx < w) s += c[(size_t)y * w + xx];
                }
                t[(size_t)y * w + x] = s;
            }
        for (int y = 0; y < h; y++)          /* vertical pass + first-wins arg-min */
            for (int x = 0; x < w; x++) {
                i32 s = 0;
                for (int k = -half; k <= half; k++) {
                    int yy = y + k;
                    if (mode == SMO_TOROIDAL) s += t[(size_t)wrap(yy, h) * w + x];
                    else if (yy >= 0 && yy < h) s += t[(size_t)yy * w + x];
                }
                if (s < best[(size_t)y * w + x]) {
                    best[(size_t)y * w + x] = s;
                    web[(size_t)y * w + x] = d + 1;
                }
            }
    }
    free(c);
    free(t);
}

/* ------------------------------------------------------------------ */
/* step 3: hole filling and contour lines                              */
/* ------------------------------------------------------------------ */

void smo_fill_web_holes(int32_t *web, int w, int h, int times)
{
    /* src/stereo.c:230-251.  Neighbours are taken at flat offsets +-1 and
     * +-w with NO wrap (IDX, not idx): x-1 at x=0 is the previous row's last
     * pixel.  Offsets that leave the array are undefined behaviour in the
     * reference and unreachable from its pipeline (the web never holds a 0,
     * SURVEY.md section 8f); here they read as 0.  The reference ping-pongs
     * two buffers by swapping pointers and returns whichever is current,
     * leaving stale values in non-hole pixels of the other one; copying the
     * current buffer back reproduces the returned image exactly. */
    const long n = (long)w * h;
    i32 *cur = zalloc(sizeof(i32) * (size_t)n); /* "web" of the reference */
    i32 *oth = zalloc(sizeof(i32) * (size_t)n); /* "tmp" */
    memcpy(cur, web, sizeof(i32) * (size_t)n);
    memcpy(oth, web, sizeof(i32) * (size_t)n);
    for (int it = 0; it < times; it++) {
        for (long p = 0; p < n; p++) {
            if (oth[p] == 0) {
                i32 r = p + 1 < n ? oth[p + 1] : 0;
                i32 u = p + w < n ? oth[p + w] : 0;
                i32 l = p - 1 >= 0 ? oth[p - 1] : 0;
                i32 d = p - w >= 0 ? oth[p - w] : 0;
                cur[p] = (r + u + l + d) / 4;
            }
        }
        i32 *t = cur;
        cur = oth;
        oth = t;
    }
    memcpy(web, cur, sizeof(i32) * (size_t)n);
    free(cur);
    free(oth);
}

int smo_draw_contour_map(const int32_t *web, int w, int h, int num_lines,
                         uint8_t *out)
{
    /* src/stereo.c:256-274 */
    const size_t n = (size_t)w * h;
    i32 lo = INT_MAX, hi = INT_MIN;
    for (size_t p = 0; p < n; p++) {
        if (web[p] < lo) lo = web[p];
        if (web[p] > hi) hi = web[p];
    }
    if (num_lines == 0)
        return -1;
    i32 interval = (hi - lo) / num_lines;
    if (interval == 0)
        return -1; /* the reference traps here (SIGFPE) */
    for (size_t p = 0; p < n; p++)
        out[p] = (u8)(((web[p] - lo) % interval) == 0);
    return 0;
}

int smo_pipeline(const uint8_t *left, const uint8_t *right, int w, int h,
                 double threshold, int num_shifts, int square_width, int times,
                 int lines, int mode, int faithful, uint8_t *edges_l,
                 uint8_t *edges_r, int32_t *best, int32_t *web1,
                 int32_t *web2, uint8_t *out)
{
    const size_t n = (size_t)w * h;
    u8 *el = edges_l ? edges_l : zalloc(n);
    u8 *er = edges_r ? edges_r : zalloc(n);
    i32 *web = zalloc(sizeof(i32) * n);
    int rc = 0;

    smo_find_all_edges(left, w, h, threshold, mode, el);
    smo_find_all_edges(right, w, h, threshold, mode, er);
    smo_hot_path(el, er, w, h, num_shifts, square_width, mode, faithful, best, web);
    if (web1)
        memcpy(web1, web, sizeof(i32) * n);
    if (web2 || out) {
        smo_fill_web_holes(web, w, h, times);
        if (web2)
            memcpy(web2, web, sizeof(i32) * n);
        if (out)
            rc = smo_draw_contour_map(web, w, h, lines, out);
    }
    if (!edges_l) free(el);
    if (!edges_r) free(er);
    free(web);
    return rc;
}
