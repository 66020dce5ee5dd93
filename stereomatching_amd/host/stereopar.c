/*
 * stereopar.c -- the GPU programs `stereopar` and (with -DGHOST)
 * `stereopar-ghost`: plain C host code over the C ABI of
 * include/stereo_hip.h.  Drop-in for the reference's CUDA programs
 * (/root/reference/src/stereo.cu, src/stereo-ghost.cu): same argv, defaults,
 * validation, messages and exit codes (src/stereo.cu:350-409), same stage
 * order and dump set (src/stereo.cu:296-347), same stdout line and timed
 * region (after allocation and upload, through the final device sync).
 *
 * Differences, all outside the contract test/diff.sh and test/time.sh check:
 *   - inputs are uploaded as uint8 (1 B/pixel) instead of double (8 B/pixel);
 *     the brightness k/256.0 is formed on the device;
 *   - the number of shifts is still compile-time NUM_SHIFTS = 30 by default but
 *     can be set at run time with the environment variable STEREO_NUM_SHIFTS;
 *   - a zero contour interval (the reference's kernel computes `% 0`) is
 *     reported on stderr with exit code 1.
 */
#include "image.h"
#include "stereo_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef NUM_SHIFTS
#define NUM_SHIFTS 30
#endif
#define DEFAULT_THRESHOLD 0.15
#define DEFAULT_SQUARE_WIDTH 21
#define DEFAULT_TIMES 32
#define DEFAULT_LINES 10

#ifdef GHOST
#define PROGRAM PARGHOST
#define BORDER SM_GHOST
#else
#define PROGRAM PAR
#define BORDER SM_TOROIDAL
#endif

typedef struct AlgorithmParams {
    double threshold;
    int square_width;
    int times;
    int lines_to_draw;
} AlgorithmParams;

/* any failure of the GPU layer: message on stderr, exit(EXIT_FAILURE), the
 * reference's checkCudaErrors convention (src/helper_cuda.h:890-901) */
#define GPU(call)                                      \
    do {                                               \
        if ((call) != SM_OK) {                         \
            fprintf(stderr, "%s\n", sm_last_error());  \
            exit(EXIT_FAILURE);                        \
        }                                              \
    } while (0)

static double get_time(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + (double)ts.tv_nsec / 1e9;
}

static void *gpu_alloc(size_t bytes)
{
    void *p = NULL;
    GPU(sm_malloc(0, bytes, &p)); /* zero-filled, like cuda_xmalloc */
    return p;
}

/* strtod/strtol with the reference's "0 and nothing consumed" error rule
 * (src/util.h:63-75) */
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

static void algorithm(const uint8_t *first, const uint8_t *second, int width, int height,
                      AlgorithmParams params, int num_shifts)
{
    const size_t n = (size_t)width * height;
    sm_plan *plan = NULL;
    GPU(sm_plan_create(0, width, height, num_shifts, params.square_width, BORDER, 1, &plan));

    uint8_t *first_edges = gpu_alloc(n), *second_edges = gpu_alloc(n), *out = gpu_alloc(n);
    int32_t *buf = gpu_alloc(n * sizeof(int32_t)), *web = gpu_alloc(n * sizeof(int32_t)),
            *tmp = gpu_alloc(n * sizeof(int32_t)), *minmax = gpu_alloc(2 * sizeof(int32_t));
#ifndef NO_WRITES
    uint8_t *match = gpu_alloc(n);
    int32_t *plane = gpu_alloc(n * sizeof(int32_t));
#endif

    /* set-up that depends on the parameters only, next to the allocations */
    GPU(sm_plan_prepare_threshold(plan, params.threshold, NULL));

    double t1 = get_time();

    /* first step: find edges in both images */
    GPU(sm_find_edges(plan, first, second, params.threshold, 1, first_edges, second_edges, NULL));
    write_gpu_image(first_edges, width, height, 0, IMTYPE_BINARY, make_filename("edges", PROGRAM, 1));
    write_gpu_image(second_edges, width, height, 0, IMTYPE_BINARY, make_filename("edges", PROGRAM, 2));

    /* second step: match edges between images.  The per-shift planes exist
     * only to be dumped; the fused launch below never materialises them. */
#ifndef NO_WRITES
    for (int i = 0; i < num_shifts; i++) {
        GPU(sm_debug_planes(plan, 0, i, match, NULL, NULL, NULL));
        write_gpu_image(match, width, height, 0, IMTYPE_BINARY, make_filename("matches", PROGRAM, i));
    }
    for (int i = 0; i < num_shifts; i++) {
        GPU(sm_debug_planes(plan, 0, i, NULL, plane, NULL, NULL));
        write_gpu_image(plane, width, height, 0, IMTYPE_GRAY_INT, make_filename("score_all", PROGRAM, i));
    }
    for (int i = 0; i < num_shifts; i++) {
        GPU(sm_debug_planes(plan, 0, i, NULL, NULL, plane, NULL));
        write_gpu_image(plane, width, height, 0, IMTYPE_GRAY_INT, make_filename("scores", PROGRAM, i));
    }
#endif
    GPU(sm_match_wta(plan, 1, web, buf, NULL));
    write_gpu_image(buf, width, height, 0, IMTYPE_GRAY_INT, make_filename("score_best", PROGRAM, 0));
    write_gpu_image(web, width, height, 0, IMTYPE_GRAY_INT, make_filename("web", PROGRAM, 1));

    /* third step: draw contour lines */
    int in_tmp = 0;
#ifdef NO_WRITES
    /* nothing is dumped between the stages: queue all of step 3, synchronise once */
    GPU(sm_step3(plan, web, tmp, params.times, params.lines_to_draw, 1, minmax, out, &in_tmp, NULL));
#else
    GPU(sm_fill_web_holes(plan, web, tmp, params.times, 1, &in_tmp, NULL));
    int32_t *filled = in_tmp ? tmp : web;
    write_gpu_image(filled, width, height, 0, IMTYPE_GRAY_INT, make_filename("web", PROGRAM, 2));
    GPU(sm_min_max(plan, filled, 1, minmax, NULL));
    GPU(sm_draw_contour_map(plan, filled, minmax, params.lines_to_draw, 1, out, NULL));
    GPU(sm_plan_status(plan, NULL)); /* synchronises; reports a zero interval */
    write_gpu_image(out, width, height, 0, IMTYPE_BINARY, make_filename("output", PROGRAM, 0));
#endif

    GPU(sm_stream_sync(0, NULL));
    double t2 = get_time();
    double elapsed = t2 - t1;
    printf("width = %d, height = %d, t1 = %f, t2 = %f, elapsed = %f\n", width, height, t1, t2, elapsed);

    GPU(sm_free(0, first_edges));
    GPU(sm_free(0, second_edges));
    GPU(sm_free(0, web));
    GPU(sm_free(0, out));
    GPU(sm_free(0, buf));
    GPU(sm_free(0, tmp));
    GPU(sm_free(0, minmax));
#ifndef NO_WRITES
    GPU(sm_free(0, match));
    GPU(sm_free(0, plane));
#endif
    sm_plan_destroy(plan);
}

/* optional positional arguments after the two images, in order */
enum { ARG_DOUBLE, ARG_INT };
static const struct {
    const char *name;   /* as it appears in "error: <name> must be a number" */
    int kind;
} OPTIONAL_ARGS[] = {{"threshold", ARG_DOUBLE}, {"square_width", ARG_INT}, {"times", ARG_INT}, {"lines", ARG_INT}};

static int fail(const char *message)
{
    fprintf(stderr, "%s\n", message);
    return 1;
}

int main(int argc, char *argv[])
{
    AlgorithmParams params = {DEFAULT_THRESHOLD, DEFAULT_SQUARE_WIDTH, DEFAULT_TIMES, DEFAULT_LINES};
    if (argc < 3) {
        fprintf(stderr, "usage: stereomatch [image 1] [image 2] [threshold = %g] "
                        "[square_width = %d] [times = %d] [lines = %d]\n",
                params.threshold, params.square_width, params.times, params.lines_to_draw);
        return 1;
    }

    uint8_t *host_image[2] = {NULL, NULL};
    int width[2], height[2];
    for (int i = 0; i < 2; i++)
        if (read_image_u8(argv[1 + i], &host_image[i], &width[i], &height[i]))
            return 1;
    if (width[0] != width[1] || height[0] != height[1])
        return fail("error: the two images must have equal width and height");

    /* same order, parse rule and messages as the reference (src/stereo.cu:371-395) */
    int *int_slot[] = {NULL, &params.square_width, &params.times, &params.lines_to_draw};
    for (int i = 0; i < 4 && 3 + i < argc; i++) {
        const int bad = OPTIONAL_ARGS[i].kind == ARG_DOUBLE ? parse_double(argv[3 + i], &params.threshold)
                                                            : parse_int(argv[3 + i], int_slot[i]);
        if (bad) {
            fprintf(stderr, "error: %s must be a number\n", OPTIONAL_ARGS[i].name);
            return 1;
        }
    }
    if (params.threshold < 0.0 || params.threshold > 1.0)
        return fail("error: threshold must be between 0 and 1");
    if (params.square_width > width[0] || params.square_width > height[0])
        return fail("error: square width must not be higher than image width/height");

    int num_shifts = NUM_SHIFTS;
    const char *env = getenv("STEREO_NUM_SHIFTS");
    if (env && atoi(env) > 0)
        num_shifts = atoi(env);

    /* upload (outside the timed region, like MAKE_GPU_COPY in src/stereo.cu:402-403) */
    const size_t n = (size_t)width[0] * height[0];
    uint8_t *device_image[2];
    for (int i = 0; i < 2; i++) {
        device_image[i] = gpu_alloc(n);
        GPU(sm_memcpy_h2d(0, device_image[i], host_image[i], n));
    }

    algorithm(device_image[0], device_image[1], width[0], height[0], params, num_shifts);

    for (int i = 0; i < 2; i++) {
        GPU(sm_free(0, device_image[i]));
        free(host_image[i]);
    }
    return 0;
}
