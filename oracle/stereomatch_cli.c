/*
 * stereomatch_cli.c -- command-line front end of the CPU oracle: the programs
 * `stereomatch` and (with -DGHOST) `stereomatch-ghost`.
 *
 * TEST INFRASTRUCTURE ONLY (see stereo_oracle.h): this is the "ser"/"sergh"
 * side of the reference's test/diff.sh, the thing the GPU programs are
 * compared against; nothing in the product links it.  Same argv, messages,
 * dump set and stdout line as /root/reference/src/stereo.c:287-392, computed
 * by the restatement in stereo_oracle.c (pinned bit-exact to the compiled
 * reference).  STEREO_NUM_SHIFTS overrides the shift count (default 30);
 * STEREO_FAITHFUL=1 selects the reference's own loop nest for the window
 * sums (slow; used to time the CPU baseline through the CLI).
 */
#include "image.h"
#include "stereo_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef NUM_SHIFTS
#define NUM_SHIFTS 30
#endif
#ifdef GHOST
#define PROGRAM SERGHOST
#define MODE SMO_GHOST
#else
#define PROGRAM SER
#define MODE SMO_TOROIDAL
#endif

static int parse_double(const char *s, double *n)
{
    char *end;
    *n = strtod(s, &end);
    return *n == 0 && end == s;
}

static int parse_int(const char *s, int *n)
{
    char *end;
    *n = (int)strtol(s, &end, 0);
    return *n == 0 && end == s;
}

static void *zalloc(size_t n)
{
    void *p = calloc(n ? n : 1, 1);
    if (!p) {
        fprintf(stderr, "error: out of memory\n");
        exit(1);
    }
    return p;
}

int main(int argc, char *argv[])
{
    double threshold = 0.15;
    int square_width = 21, times = 32, lines = 10;
    if (argc < 3) {
        fprintf(stderr, "usage: stereomatch [image 1] [image 2] [threshold = %g] "
                        "[square_width = %d] [times = %d] [lines = %d]\n",
                threshold, square_width, times, lines);
        return 1;
    }
    uint8_t *left = NULL, *right = NULL;
    int w, h, w2, h2;
    if (read_image_u8(argv[1], &left, &w, &h) || read_image_u8(argv[2], &right, &w2, &h2))
        return 1;
    if (w != w2 || h != h2) {
        fprintf(stderr, "error: the two images must have equal width and height\n");
        return 1;
    }
    static const char *const NAMES[] = {"threshold", "square_width", "times", "lines"};
    int *ints[] = {NULL, &square_width, &times, &lines};
    for (int i = 0; i < 4 && 3 + i < argc; i++) {
        if (i == 0 ? parse_double(argv[3], &threshold) : parse_int(argv[3 + i], ints[i])) {
            fprintf(stderr, "error: %s must be a number\n", NAMES[i]);
            return 1;
        }
    }
    if (threshold < 0.0 || threshold > 1.0) {
        fprintf(stderr, "error: threshold must be between 0 and 1\n");
        return 1;
    }
    if (square_width > w || square_width > h) {
        fprintf(stderr, "error: square width must not be higher than image width/height\n");
        return 1;
    }
    int num_shifts = NUM_SHIFTS;
    if (getenv("STEREO_NUM_SHIFTS") && atoi(getenv("STEREO_NUM_SHIFTS")) > 0)
        num_shifts = atoi(getenv("STEREO_NUM_SHIFTS"));
    const int faithful = getenv("STEREO_FAITHFUL") && atoi(getenv("STEREO_FAITHFUL"));

    const size_t n = (size_t)w * h;
    uint8_t *e1 = zalloc(n), *e2 = zalloc(n), *out = zalloc(n), *match = zalloc(n);
    int32_t *sum = zalloc(4 * n), *score = zalloc(4 * n), *best = zalloc(4 * n), *web = zalloc(4 * n);

    double t1 = smo_time();
    smo_find_all_edges(left, w, h, threshold, MODE, e1);
    smo_find_all_edges(right, w, h, threshold, MODE, e2);
    write_image(e1, w, h, 0, IMTYPE_BINARY, make_filename("edges", PROGRAM, 1));
    write_image(e2, w, h, 0, IMTYPE_BINARY, make_filename("edges", PROGRAM, 2));
#ifndef NO_WRITES
    /* the per-shift planes, in the reference's dump order */
    for (int i = 0; i < num_shifts; i++) {
        smo_match_plane(e1, e2, w, h, i, MODE, match);
        write_image(match, w, h, 0, IMTYPE_BINARY, make_filename("matches", PROGRAM, i));
    }
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < num_shifts; i++) {
            smo_match_plane(e1, e2, w, h, i, MODE, match);
            memset(sum, 0, 4 * n);
            (faithful ? smo_addup_faithful : smo_addup_fast)(match, w, h, square_width, MODE, sum);
            if (pass == 0) {
                write_image(sum, w, h, 0, IMTYPE_GRAY_INT, make_filename("score_all", PROGRAM, i));
            } else {
                memset(score, 0, 4 * n);
                smo_record_score(match, sum, w, h, score);
                write_image(score, w, h, 0, IMTYPE_GRAY_INT, make_filename("scores", PROGRAM, i));
            }
        }
    }
#endif
    smo_hot_path(e1, e2, w, h, num_shifts, square_width, MODE, faithful, best, web);
    write_image(best, w, h, 0, IMTYPE_GRAY_INT, make_filename("score_best", PROGRAM, 0));
    write_image(web, w, h, 0, IMTYPE_GRAY_INT, make_filename("web", PROGRAM, 1));
    smo_fill_web_holes(web, w, h, times);
    write_image(web, w, h, 0, IMTYPE_GRAY_INT, make_filename("web", PROGRAM, 2));
    if (smo_draw_contour_map(web, w, h, lines, out)) {
        /* the reference divides by the zero interval here */
        volatile int zero = 0;
        return 1 / zero;
    }
    write_image(out, w, h, 0, IMTYPE_BINARY, make_filename("output", PROGRAM, 0));
    double t2 = smo_time();
    printf("width = %d, height = %d, t1 = %f, t2 = %f, elapsed = %f\n", w, h, t1, t2, t2 - t1);
    return 0;
}
