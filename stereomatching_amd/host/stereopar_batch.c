/*
 * stereopar_batch.c -- `stereopar-batch`: a BATCH of independent stereo pairs
 * over every visible GPU, plain C host code over the C ABI of
 * include/stereo_hip.h.
 *
 * The reference processes one pair on one device (src/stereo.cu:350-409:
 * main() reads two images, uploads them, runs algorithm() on device 0).  This
 * program keeps that single-pair contract per pair -- same decoding, same
 * parameters and validation, same stages up to the web map -- and adds what
 * the reference does not have: pair j goes to device j mod n_devices (no
 * exchange between devices: pairs are independent), each device has its own
 * host thread, plan and streams; the images are decoded into one pinned arena
 * and uploaded straight from it; uploads, kernels and downloads run on a stream
 * each (a DMA queue per copy direction: both directions of the link are busy at
 * once), chained with events over three buffer sets, so that the upload of batch
 * k+1, the kernels of batch k and the download of batch k-1 overlap.  Results
 * come back as NARROW web maps (uint8 when the shifts fit,
 * else uint16: sm_run_typed), a quarter / half of the int32 PCIe traffic.
 *
 *   stereopar-batch [options] LIST [threshold] [square_width]
 *     LIST          text file, one pair per line: "left-image right-image"
 *                   (8-bit gray PNG or binary PGM, all pairs the same size)
 *     threshold, square_width   as for stereopar (defaults 0.15, 21)
 *   options
 *     -d 0,1,...    devices to use (default: $STEREO_DEVICES, else all visible)
 *     -n SHIFTS     number of shifts (default: $STEREO_NUM_SHIFTS, else 30)
 *     -g            ghost borders (stereopar-ghost semantics) instead of toroidal
 *     -b PAIRS      pairs per launch (default 8)
 *     -o DIR        write each web map to DIR/web-<index>.pgm (binary PGM, maxval =
 *                   shifts; DIR must exist).  Default: no files (timing).
 *     -r REPEAT     process the list REPEAT times (throughput measurements)
 *   builds with -DSTEREOPAR_BATCH_TEST_HOOKS (the test and ThreadSanitizer builds, never the product) add
 *     -x N          every device reports a failure when it is about to submit its N-th batch
 *                   (exercises the error path of the two host threads per device)
 *
 * stdout: one line
 *   pairs = P, devices = K, width = W, height = H, shifts = D, elapsed = T, pairs_per_s = R, checksum = C
 * where elapsed covers upload, kernels and download of all pairs (decoding is
 * done before the clock starts, as the reference reads its images before t1)
 * and checksum is the sum over all web pixels of all pairs (order-independent).
 * Errors: message on stderr, exit code 1, as the reference's programs.
 */
#include "image.h"
#include "stereo_hip.h"
#include "batch_index.h"

#include <pthread.h>
#include <stdatomic.h>
#include <semaphore.h>
#ifdef __SSE2__
#include <emmintrin.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef NUM_SHIFTS
#define NUM_SHIFTS 30
#endif
#define DEFAULT_THRESHOLD 0.15
#define DEFAULT_SQUARE_WIDTH 21
#define MAX_DEVICES 64
#define SLOTS 3

typedef struct {
    char *left, *right;
    uint8_t *px[2];        /* decoded samples, in the pinned arena (see main) */
} Pair;

typedef struct {
    /* job */
    int device, n_devices, rank;      /* rank = position of this device in the device list */
    Pair *pairs;
    int n_pairs, repeat;
    int width, height, num_shifts, square_width, border, batch;
    double threshold;
    const char *out_dir;
    pthread_barrier_t *start;
    /* result */
    unsigned long long checksum;
    int done_pairs;
    atomic_int failed;        /* set by either of the device's two host threads, read by both */
    char error[512];          /* written by the thread that sets `failed` first (error_claimed) */
    atomic_int error_claimed;
    int fail_at;              /* -x: inject a failure at this batch (0 = never) */
    /* the device's two host threads: the submitter (worker_main) queues batches into SLOTS
     * buffer sets; the collector (collector_main) waits for each download, sums and writes the
     * maps and hands the set back -- reading 2 MB per map takes the host longer than the GPU
     * and the link need for it, so it must not sit between two submissions */
    sem_t slot_free, slot_filled;
    /* A set's in_flight / first_index are written by the submitter BEFORE it posts slot_filled for
     * that set and read by the collector AFTER it has waited for it (and the other way round with
     * slot_free): the semaphores order them.  The end is signalled separately -- `submitted` =
     * batches handed over so far, `finished` = no more will come -- never by rewriting the fields
     * of a set the collector may be looking at. */
    int in_flight[SLOTS];                         /* pairs in the set */
    long first_index[SLOTS];                      /* the first pair's index in this device's sequence (mine * repeat
                                                   * can pass 2^31) */
    atomic_long submitted;
    atomic_int finished;
    void *ev_down[SLOTS], *h_web[SLOTS];
    int mine, web_bytes;
} Worker;

static double get_time(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + (double)ts.tv_nsec / 1e9;
}

static int parse_double(const char *s, double *n)   /* src/util.h:63-75 */
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

static int fail(const char *message)
{
    fprintf(stderr, "%s\n", message);
    return 1;
}

/* the first thread to report a failure owns the message */
static void set_failed(Worker *w, const char *message)
{
    if (atomic_exchange(&w->error_claimed, 1) == 0)
        snprintf(w->error, sizeof w->error, "%s", message);
    atomic_store(&w->failed, 1);
}

#define W_TRY(call)                                                        \
    do {                                                                   \
        if ((call) != SM_OK) {                                             \
            set_failed(w, sm_last_error());                                \
            goto out;                                                      \
        }                                                                  \
    } while (0)

static void write_pgm(const char *dir, int index, const void *web, int bytes, int width, int height,
                      int maxval)
{
    char name[1200];
    snprintf(name, sizeof name, "%s/web-%d.pgm", dir, index);
    FILE *f = fopen(name, "wb");
    if (!f)
        return;                 /* like write_image: silently skip what cannot be opened */
    fprintf(f, "P5\n%d %d\n%d\n", width, height, maxval);
    const size_t n = (size_t)width * height;
    if (bytes == 1) {
        fwrite(web, 1, n, f);
    } else {                    /* 16-bit PGM is big-endian */
        const uint16_t *v = web;
        for (size_t i = 0; i < n; i++) {
            fputc(v[i] >> 8, f);
            fputc(v[i] & 255, f);
        }
    }
    fclose(f);
}

/* sum of n bytes, eight at a time: the even and the odd bytes of a 64-bit word are added into
 * four 16-bit lanes (each step adds at most 510 to a lane, so 128 steps fit), which are
 * folded into the total before they can overflow.  The map sums are on the host thread's
 * critical path between two batches. */
static unsigned long long sum_bytes(const uint8_t *p, size_t n)
{
    const unsigned long long M = 0x00ff00ff00ff00ffULL;
    unsigned long long total = 0;
    size_t i = 0;
#ifdef __SSE2__
    /* x86-64: psadbw against zero sums 16 bytes per instruction into two 64-bit lanes */
    {
        __m128i acc = _mm_setzero_si128();
        const __m128i zero = _mm_setzero_si128();
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(p + i)), zero);
            const __m128i b = _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(p + i + 16)), zero);
            const __m128i c = _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(p + i + 32)), zero);
            const __m128i d = _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(p + i + 48)), zero);
            acc = _mm_add_epi64(acc, _mm_add_epi64(_mm_add_epi64(a, b), _mm_add_epi64(c, d)));
        }
        unsigned long long lanes[2];
        _mm_storeu_si128((__m128i *)lanes, acc);
        total = lanes[0] + lanes[1];
    }
#endif
    while (i + 8 <= n) {
        unsigned long long lanes = 0;
        size_t steps = (n - i) / 8;
        if (steps > 128) steps = 128;
        for (size_t k = 0; k < steps; k++, i += 8) {
            unsigned long long w;
            memcpy(&w, p + i, 8);
            lanes += (w & M) + ((w >> 8) & M);
        }
        total += (lanes & 0xffff) + ((lanes >> 16) & 0xffff) + ((lanes >> 32) & 0xffff) + (lanes >> 48);
    }
    for (; i < n; i++) total += p[i];
    return total;
}

/* the collector of one device: batches come back in the order they were submitted */
static void *collector_main(void *arg)
{
    Worker *w = arg;
    const size_t n = (size_t)w->width * w->height;
    long processed = 0;
    for (int s = 0;; s = (s + 1) % SLOTS, processed++) {
        sem_wait(&w->slot_filled);
        /* one post per submitted batch, and one more when the submitter is done or has failed */
        if (atomic_load(&w->failed))
            break;
        if (processed == atomic_load(&w->submitted)) {
            /* a post with no batch behind it: the end signal (`finished` is set before that post) */
            if (!atomic_load(&w->finished))
                set_failed(w, "error: collector woken without a batch or the end signal");
            break;
        }
        const int b = w->in_flight[s];
        if (sm_event_sync(w->device, w->ev_down[s]) != SM_OK)
            set_failed(w, sm_last_error());
        for (int k = 0; k < b && !atomic_load(&w->failed); k++) {
            const uint8_t *m8 = (const uint8_t *)w->h_web[s] + (size_t)k * n * w->web_bytes;
            unsigned long long sum = 0;
            if (w->web_bytes == 1) {
                sum = sum_bytes(m8, n);
            } else {            /* little-endian uint16: low bytes + 256 * high bytes */
                const uint16_t *m16 = (const uint16_t *)m8;
                for (size_t i = 0; i < n; i++) sum += m16[i];
            }
            w->checksum += sum;
            const long seq = w->first_index[s] + k;                         /* in this device's sequence */
            const int j = batch_global_index(w->rank, seq, w->mine, w->n_devices);   /* line of the pair list */
            if (w->out_dir && seq < w->mine)
                write_pgm(w->out_dir, j, m8, w->web_bytes, w->width, w->height, w->num_shifts);
            w->done_pairs++;
        }
        sem_post(&w->slot_free);
    }
    return NULL;
}

/* one device: its share of the pairs; three buffer sets in flight over three streams */
static void *worker_main(void *arg)
{
    Worker *w = arg;
    const int dev = w->device;
    const size_t n = (size_t)w->width * w->height;
    const int web_type = w->num_shifts <= 255 ? SM_WEB_U8 : SM_WEB_U16;
    const int web_bytes = web_type == SM_WEB_U8 ? 1 : 2;
    sm_plan *plan = NULL;                 /* all kernels run on ONE stream, in order: one plan */
    void *st_up = NULL, *st_run = NULL, *st_down = NULL;
    void *ev_up[SLOTS] = {NULL}, *ev_ran[SLOTS] = {NULL};
    uint8_t *d_in[SLOTS] = {NULL};
    void *d_web[SLOTS] = {NULL};
    int used[SLOTS] = {0};
    pthread_t collector;
    int collector_started = 0;

    /* this device's pairs: j with j mod n_devices == rank, over all repeats */
    const int mine = batch_pairs_of_rank(w->n_pairs, w->rank, w->n_devices);
    const long total = (long)mine * w->repeat;
    w->mine = mine;
    w->web_bytes = web_bytes;
    sem_init(&w->slot_free, 0, SLOTS);
    sem_init(&w->slot_filled, 0, 0);

    /* allocation and set-up, before the clock starts (src/stereo.cu:296-308) */
    if (sm_plan_create(dev, w->width, w->height, w->num_shifts, w->square_width, w->border,
                       w->batch, &plan) ||
        sm_plan_prepare_threshold(plan, w->threshold, NULL) ||
        sm_stream_create(dev, &st_up) || sm_stream_create(dev, &st_run) || sm_stream_create(dev, &st_down))
        set_failed(w, sm_last_error());
    for (int s = 0; s < SLOTS && !atomic_load(&w->failed); s++) {
        if (sm_event_create(dev, &ev_up[s]) || sm_event_create(dev, &ev_ran[s]) ||
            sm_event_create(dev, &w->ev_down[s]) ||
            sm_host_alloc(n * web_bytes * w->batch, &w->h_web[s]) ||
            sm_malloc(dev, 2 * n * w->batch, (void **)&d_in[s]) ||
            sm_malloc(dev, n * web_bytes * w->batch, &d_web[s]))
            set_failed(w, sm_last_error());
    }
    /* one pair through every stream before the clock starts: the first transfer and the first
     * launch on a stream set up DMA queues and load code objects (tens of milliseconds) */
    if (!atomic_load(&w->failed) && mine > 0) {
        const int j = w->rank;
        if (sm_memcpy_h2d_async(dev, d_in[0], w->pairs[j].px[0], n, st_up) ||
            sm_memcpy_h2d_async(dev, d_in[0] + n, w->pairs[j].px[1], n, st_up) ||
            sm_stream_sync(dev, st_up) ||
            sm_run_typed(plan, d_in[0], d_in[0] + n, w->threshold, 1, d_web[0], web_type, NULL, st_run) ||
            sm_stream_sync(dev, st_run) ||
            sm_memcpy_d2h_async(dev, w->h_web[0], d_web[0], n * web_bytes, st_down) ||
            sm_stream_sync(dev, st_down))
            set_failed(w, sm_last_error());
    }
    if (!atomic_load(&w->failed)) {
        if (pthread_create(&collector, NULL, collector_main, w) == 0)
            collector_started = 1;
        else
            set_failed(w, "error: cannot start the collector thread");
    }
    pthread_barrier_wait(w->start);       /* main() takes t1 here */
    if (atomic_load(&w->failed))
        goto out;

    long next = 0, batches = 0;           /* index into this device's sequence of pairs; batches submitted */
    for (int s = 0; next < total && !atomic_load(&w->failed); s = (s + 1) % SLOTS) {
        sem_wait(&w->slot_free);          /* the collector is done with this set's previous maps */
#ifdef STEREOPAR_BATCH_TEST_HOOKS
        if (w->fail_at && batches + 1 == w->fail_at) {
            set_failed(w, "error: injected failure (-x)");
            goto out;
        }
#endif
        /* upload the next batch straight from the pinned arena the images were decoded
         * into: lefts then rights, as sm_run expects a batch (no staging copy on the host).
         * The set's input buffer is free once the kernels that last read it have run,
         * its map buffer once the download that last read it has finished. */
        int b = 0;
        while (b < w->batch && next + b < total) b++;
        if (used[s]) W_TRY(sm_stream_wait_event(dev, st_up, ev_ran[s]));
        for (int k = 0; k < b; k++) {
            const int j = batch_global_index(w->rank, next + k, mine, w->n_devices);
            W_TRY(sm_memcpy_h2d_async(dev, d_in[s] + (size_t)k * n, w->pairs[j].px[0], n, st_up));
            W_TRY(sm_memcpy_h2d_async(dev, d_in[s] + (size_t)(b + k) * n, w->pairs[j].px[1], n, st_up));
        }
        W_TRY(sm_event_record(dev, ev_up[s], st_up));
        /* The batch's kernels wait for ITS upload and for nothing else on st_run (sm_run_after: the event is the call's
         * only input dependency; round 5) -- the edge detection of this batch may then run beside the match launch of the
         * one before where that launch leaves room (batches of few small pairs).  The set's map buffer is free: the
         * collector has waited for the download that last read it (ev_down) before it posted slot_free. */
        W_TRY(sm_run_after(plan, d_in[s], d_in[s] + (size_t)b * n, w->threshold, b, d_web[s],
                           web_type, NULL, st_run, ev_up[s]));
        W_TRY(sm_event_record(dev, ev_ran[s], st_run));
        W_TRY(sm_stream_wait_event(dev, st_down, ev_ran[s]));
        W_TRY(sm_memcpy_d2h_async(dev, w->h_web[s], d_web[s], n * web_bytes * b, st_down));
        W_TRY(sm_event_record(dev, w->ev_down[s], st_down));
        used[s] = 1;
        w->first_index[s] = next;
        w->in_flight[s] = b;
        next += b;
        atomic_store(&w->submitted, ++batches);
        sem_post(&w->slot_filled);
    }
out:
    if (collector_started) {
        /* the end, successful or not: one more post.  The collector takes the batches in
         * submission order and stops when it has seen `submitted` of them (or a failure). */
        atomic_store(&w->finished, 1);
        sem_post(&w->slot_filled);
        pthread_join(collector, NULL);
    }
    if (st_up) sm_stream_sync(dev, st_up);
    if (st_run) sm_stream_sync(dev, st_run);
    if (st_down) sm_stream_sync(dev, st_down);
    for (int k = 0; k < SLOTS; k++) {
        if (ev_up[k]) sm_event_destroy(dev, ev_up[k]);
        if (ev_ran[k]) sm_event_destroy(dev, ev_ran[k]);
        if (w->ev_down[k]) sm_event_destroy(dev, w->ev_down[k]);
        if (w->h_web[k]) sm_host_free(w->h_web[k]);
        if (d_in[k]) sm_free(dev, d_in[k]);
        if (d_web[k]) sm_free(dev, d_web[k]);
    }
    if (st_up) sm_stream_destroy(dev, st_up);
    if (st_run) sm_stream_destroy(dev, st_run);
    if (st_down) sm_stream_destroy(dev, st_down);
    if (plan) sm_plan_destroy(plan);
    sem_destroy(&w->slot_free);
    sem_destroy(&w->slot_filled);
    return NULL;
}

static int parse_devices(const char *list, int *devices, int visible)
{
    int n = 0;
    char *copy = strdup(list), *save = NULL;
    for (char *tok = strtok_r(copy, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
        int d;
        if (parse_int(tok, &d) || d < 0 || d >= visible || n == MAX_DEVICES) {
            free(copy);
            return -1;
        }
        devices[n++] = d;
    }
    free(copy);
    return n;
}

int main(int argc, char *argv[])
{
    const char *device_list = getenv("STEREO_DEVICES"), *out_dir = NULL;
    int num_shifts = NUM_SHIFTS, border = SM_TOROIDAL, batch = 8, repeat = 1, fail_at = 0;
    const char *env = getenv("STEREO_NUM_SHIFTS");
    if (env && atoi(env) > 0)
        num_shifts = atoi(env);

    int a = 1;
    for (; a < argc && argv[a][0] == '-' && argv[a][1]; a++) {
        const char opt = argv[a][1];
        if (opt == 'g') { border = SM_GHOST; continue; }
#ifdef STEREOPAR_BATCH_TEST_HOOKS
        const char *value_options = "dnborx";
#else
        const char *value_options = "dnbor";
#endif
        if (a + 1 >= argc || !strchr(value_options, opt)) { a = argc; break; }
        const char *val = argv[++a];
        if (opt == 'd') device_list = val;
        else if (opt == 'o') out_dir = val;
        else {
            int v;
            if (parse_int(val, &v) || v < 1) {
                fprintf(stderr, "error: -%c must be a positive number\n", opt);
                return 1;
            }
            if (opt == 'n') num_shifts = v;
            if (opt == 'b') batch = v;
            if (opt == 'r') repeat = v;
            if (opt == 'x') fail_at = v;
        }
    }
    if (a >= argc) {
        fprintf(stderr, "usage: stereopar-batch [-d devices] [-n shifts] [-g] [-b pairs per launch] "
                        "[-o dir] [-r repeat] [pair list] [threshold = %g] [square_width = %d]\n",
                DEFAULT_THRESHOLD, DEFAULT_SQUARE_WIDTH);
        return 1;
    }
    const char *list_name = argv[a++];
    double threshold = DEFAULT_THRESHOLD;
    int square_width = DEFAULT_SQUARE_WIDTH;
    if (a < argc && parse_double(argv[a++], &threshold))
        return fail("error: threshold must be a number");
    if (a < argc && parse_int(argv[a++], &square_width))
        return fail("error: square_width must be a number");
    if (threshold < 0.0 || threshold > 1.0)
        return fail("error: threshold must be between 0 and 1");
    if (num_shifts > 65535)
        return fail("error: the number of shifts must not exceed 65535");

    /* the list */
    FILE *lf = fopen(list_name, "r");
    if (!lf) {
        fprintf(stderr, "error reading pair list %s:", list_name);
        perror("");
        return 1;
    }
    Pair *pairs = NULL;
    int n_pairs = 0, cap = 0;
    char l[1024], r[1024], line[2200];
    while (fgets(line, sizeof line, lf)) {
        if (line[0] == '#' || line[0] == '\n')
            continue;
        if (sscanf(line, "%1023s %1023s", l, r) != 2) {
            fprintf(stderr, "error: line %d of %s does not name two images\n", n_pairs + 1, list_name);
            return 1;
        }
        if (n_pairs == cap) {
            cap = cap ? 2 * cap : 64;
            pairs = realloc(pairs, (size_t)cap * sizeof *pairs);
            if (!pairs)
                return fail("error: out of memory");
        }
        pairs[n_pairs].left = strdup(l);
        pairs[n_pairs].right = strdup(r);
        n_pairs++;
    }
    fclose(lf);
    if (n_pairs == 0)
        return fail("error: the pair list is empty");

    /* decode (before the clock starts, as the reference reads its images before t1) into ONE
     * pinned arena, so that the uploads need no staging copy */
    int width = 0, height = 0;
    uint8_t *arena = NULL;
    for (int j = 0; j < n_pairs; j++) {
        for (int side = 0; side < 2; side++) {
            int wd, ht;
            uint8_t *px = NULL;
            if (read_image_u8(side ? pairs[j].right : pairs[j].left, &px, &wd, &ht))
                return 1;
            if (j == 0 && side == 0) {
                width = wd; height = ht;
                if (sm_host_alloc((size_t)2 * n_pairs * wd * ht, (void **)&arena) != SM_OK)
                    return fail(sm_last_error());
            }
            if (wd != width || ht != height)
                return fail("error: the two images must have equal width and height");
            pairs[j].px[side] = arena + ((size_t)2 * j + side) * wd * ht;
            memcpy(pairs[j].px[side], px, (size_t)wd * ht);
            free(px);
        }
    }
    if (square_width > width || square_width > height)
        return fail("error: square width must not be higher than image width/height");

    /* devices */
    int visible = 0;
    if (sm_device_count(&visible) != SM_OK)
        return fail(sm_last_error());
    if (visible < 1)
        return fail("error: no GPU visible");
    int devices[MAX_DEVICES], n_devices = 0;
    if (device_list && *device_list) {
        n_devices = parse_devices(device_list, devices, visible);
        if (n_devices < 1) {
            fprintf(stderr, "error: device list \"%s\" is not a list of devices 0..%d\n", device_list, visible - 1);
            return 1;
        }
    } else {
        for (; n_devices < visible && n_devices < MAX_DEVICES; n_devices++) devices[n_devices] = n_devices;
    }
    if (n_devices > n_pairs)
        n_devices = n_pairs;

    Worker workers[MAX_DEVICES];
    pthread_t threads[MAX_DEVICES];
    memset(workers, 0, sizeof workers);
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)n_devices + 1);
    for (int k = 0; k < n_devices; k++) {
        Worker *w = &workers[k];
        w->device = devices[k]; w->n_devices = n_devices; w->rank = k;
        w->pairs = pairs; w->n_pairs = n_pairs; w->repeat = repeat;
        w->width = width; w->height = height; w->num_shifts = num_shifts;
        w->square_width = square_width; w->border = border; w->batch = batch;
        w->threshold = threshold; w->out_dir = out_dir; w->start = &start; w->fail_at = fail_at;
        /* (returning from main ends the process, and with it the device threads already waiting
         * at the start barrier) */
        if (pthread_create(&threads[k], NULL, worker_main, w))
            return fail("error: cannot start a device thread");
    }
    pthread_barrier_wait(&start);         /* every device has its plans and buffers */
    const double t1 = get_time();
    unsigned long long checksum = 0;
    int done = 0, failed = 0;
    for (int k = 0; k < n_devices; k++) {
        pthread_join(threads[k], NULL);
        if (atomic_load(&workers[k].failed)) {
            fprintf(stderr, "%s\n", workers[k].error);
            failed = 1;
        }
        checksum += workers[k].checksum;
        done += workers[k].done_pairs;
    }
    const double elapsed = get_time() - t1;
    if (failed)
        return 1;
    printf("pairs = %d, devices = %d, width = %d, height = %d, shifts = %d, elapsed = %f, "
           "pairs_per_s = %f, checksum = %llu\n",
           done, n_devices, width, height, num_shifts, elapsed, done / elapsed, checksum);
    for (int j = 0; j < n_pairs; j++) {
        free(pairs[j].left); free(pairs[j].right);
    }
    free(pairs);
    sm_host_free(arena);
    return 0;
}
