/*
 * stereo_hip.h -- C ABI of the MI355X (gfx950) stereo-matching layer.
 *
 * This is the drop-in boundary for the reference's one data-parallel hot path
 * (per-shift match cost -> S x S window aggregation -> winner-take-all) plus
 * the stages either side of it.  Plain C: opaque plan handle, raw pointers,
 * sizes and a stream handle (a hipStream_t passed as void *, NULL = default
 * stream).  No C++ or torch types cross it.  The reference has no FFI of its
 * own: its boundary is main()/algorithm() calling file-local kernels, so each
 * entry point names the reference code it replaces (paths relative to
 * /root/reference).  INTEGRATION.md shows the reference-side call sites.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero (an SM_ERR_* code) on
 *     failure; sm_last_error() then returns a message for the calling thread.
 *     The reference's convention on any GPU API failure is "message on
 *     stderr, exit(EXIT_FAILURE)" (src/helper_cuda.h:890-901); callers keep
 *     that by printing sm_last_error() and exiting 1.
 *   - d_* pointers are device (HBM) addresses owned by the caller; the plan
 *     owns only its private workspace.
 *   - launches are asynchronous on the given stream; nothing here
 *     synchronises except sm_stream_sync, sm_memcpy_* and sm_plan_status.
 *   - a plan is a single-threaded, single-stream object: its workspace (the packed edge images,
 *     the staging map of the narrow results of the fallback kernels, the timing events) is shared
 *     by all calls on it, so calls on ONE plan must come from one thread at a time and their
 *     launches must be ordered (one stream, or streams chained with events).  Concurrency comes
 *     from several plans (stereopar-batch: one per device thread), not from sharing one.
 *   - images are row-major, W*H elements, no padding.  A batch of pairs is
 *     `pairs` consecutive images.  The ghost-border variant needs no padded
 *     arrays (src/ghost.h): the halo is synthesised inside the kernels.
 */
#ifndef STEREO_HIP_H
#define STEREO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_OK            0
#define SM_ERR_ARG       1 /* invalid argument */
#define SM_ERR_HIP       2 /* a HIP runtime call or a launch failed */
#define SM_ERR_NOMEM     3 /* device or host allocation failed */
#define SM_ERR_ZERO_DIV  4 /* contour interval is 0 (the reference traps: SIGFPE) */

/* border handling: stereo.c (wrap-around idx(), src/util.h:42-47) or
 * stereo-ghost.c (zero / 128.0 halos, src/stereo-ghost.c:286-287,:384-385) */
enum sm_border { SM_TOROIDAL = 0, SM_GHOST = 1 };

typedef struct sm_plan sm_plan;

/* message for the last failure on this thread ("" if none) */
const char *sm_last_error(void);

/* number of visible HIP devices */
int sm_device_count(int *count);

/* ---- device memory helpers (what ALLOCATE_GPU / MAKE_GPU_COPY /
 * MAKE_HOST_COPY of src/util.h:119-152 did; buffers come back zero-filled
 * like cuda_xmalloc, src/util.h:119-129) -------------------------------- */
int sm_malloc(int device, size_t bytes, void **d_ptr);
int sm_free(int device, void *d_ptr);
int sm_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
int sm_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
int sm_stream_sync(int device, void *stream);

/* pinned (page-locked) host buffers and asynchronous copies on a stream, for
 * callers that feed pairs from host memory and want the transfers to overlap
 * the kernels of neighbouring pairs (the reference copies synchronously from
 * pageable memory, src/util.h:131-143).  sm_stream_create/destroy hand out
 * plain hipStream_t handles for callers without a HIP runtime of their own.  */
int sm_host_alloc(size_t bytes, void **h_ptr);
int sm_host_free(void *h_ptr);
int sm_memcpy_h2d_async(int device, void *d_dst, const void *h_src, size_t bytes, void *stream);
int sm_memcpy_d2h_async(int device, void *h_dst, const void *d_src, size_t bytes, void *stream);
int sm_stream_create(int device, void **stream);
int sm_stream_destroy(int device, void *stream);
/* events order work ACROSS streams: a host that keeps uploads, kernels and downloads on three
 * streams of their own (each copy direction then has a DMA queue to itself and the two
 * directions of the link run at the same time) chains them with these.  sm_event_sync blocks
 * the calling thread until the recorded work has finished.                                  */
int sm_event_create(int device, void **event);
int sm_event_destroy(int device, void *event);
int sm_event_record(int device, void *event, void *stream);
int sm_stream_wait_event(int device, void *stream, void *event);
int sm_event_sync(int device, void *event);

/* ---- result collection across the GPUs of one node: RCCL over xGMI ------- *
 * New work (the reference drives one GPU and copies its map to the host: src/stereo.cu:402-403,
 * src/image.cu:15-23).  The hot path shards with no data-path collective -- pair j -> device j mod n,
 * SURVEY.md 8e -- and these calls are what BASELINE.json's north star names RCCL for: a broadcast of a
 * parameter block and the collection of the result maps on one GPU.  ONE process drives the n devices
 * (ncclCommInitAll; each call below enqueues the operations of all ranks inside one ncclGroupStart /
 * ncclGroupEnd); rank r is devices[r], the root is rank 0.  All calls are asynchronous on the streams
 * given (streams[r] on device r; NULL = every default stream).  librccl.so is loaded by sm_comm_create,
 * not with this library.  (A host that wants its maps in HOST memory is better served by one download
 * per device -- n PCIe links instead of one -- which is what host/stereopar_batch.c does.)            */
typedef struct sm_comm sm_comm;
/* The RCCL build to load instead of the default names (librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1): a path,
 * before the first sm_comm_create of the process.  A library that exports the symbol sm_rccl_host_stand_in is taken
 * for a HOST stand-in (tests/rccl_stub.c): its communicators are driven with host buffers and no device is touched,
 * which is how the grouped send / receive order of sm_gather_maps is rehearsed where fewer than two GPUs exist.      */
int sm_comm_set_rccl_library(const char *path);
int sm_comm_create(const int *devices, int n, sm_comm **out);     /* every device at most once */
void sm_comm_destroy(sm_comm *comm);
int sm_comm_size(const sm_comm *comm);
/* bytes at d_buf[0] (device 0 of the communicator) -> d_buf[r] on every device r */
int sm_broadcast(sm_comm *comm, void *const *d_buf, size_t bytes, void *const *streams);
/* rank r contributes bytes[r] bytes at d_src[r]; the root receives them back to back, in rank order, at
 * d_dst (its own share is a copy on its device; ranks with bytes[r] == 0 take no part): point-to-point
 * ncclSend / ncclRecv, all in flight together.  The caller maps rank order to pair order (pair j -> rank
 * j mod n: host/batch_index.h).                                                                       */
int sm_gather_maps(sm_comm *comm, void *const *d_src, const size_t *bytes, void *d_dst, void *const *streams);

/* ---- plan: geometry + private workspace for one image size ------------- *
 * num_shifts  = the reference's compile-time NUM_SHIFTS (src/stereo.c:6),
 *               here a run-time value, 1..65535
 * square_width = argv[4] of the reference (src/stereo.c:365); the window is
 *               (2*(square_width/2)+1)^2; 0 <= square_width <= min(W,H)
 *               (src/stereo.c:382-385)
 * max_pairs   = largest batch a single call will be given (>= 1)          */
int sm_plan_create(int device, int width, int height, int num_shifts,
                   int square_width, int border, int max_pairs, sm_plan **out);
void sm_plan_destroy(sm_plan *plan);

/* The same with an explicit choice of kernel variant and tiling: what tuning runs, same-device
 * A/B measurements and the tests that walk every built kernel need.  Every field 0 (or a NULL
 * `options`) = the plan's own choice, which is what sm_plan_create takes; a choice that does not
 * apply to the kernel the plan uses is ignored (sm_plan_describe / sm_plan_geometry say what was
 * taken).  The library reads no environment variable.                                      */
typedef struct sm_plan_options {
    int struct_size;            /* sizeof(sm_plan_options) as the caller compiled it: a shorter (older) struct leaves the
                                 * missing fields 0, of a longer (newer) one the fields this library knows are taken;
                                 * 0 is read as "no field set" (so `sm_plan_options o = {0}` is valid) */
    int kernel_family;          /* 1 = the popcount kernels (general fallback) even where the bit-sliced one is built */
    int tile_h;                 /* match kernel: output rows per wave */
    int shifts_per_lane;        /* bit-sliced kernel: 4, 8 or 16 */
    int workgroup_waves;        /* bit-sliced kernel: 1 = one-wave workgroups, 2 = two-wave workgroups (shared warm-up) */
    int no_two_wave_cap;        /* 1 = never launch the variant capped at two waves per SIMD */
    unsigned priority_pattern;  /* bit-sliced kernel: time-sliced wave priority, bit k = favoured slot parity in unit k */
    int edge_kernel;            /* 1 = the one-pixel-per-lane edge kernel even where the four-pixel one applies */
    int timing_by_records;      /* 1 = sm_plan_time_kernels brackets launches with event records instead of
                                 *     reading the dispatch's own time stamps */
    int cost_pixels_per_lane;   /* SAD kernel (sm_cost_wta): 2 or 4 where both are built */
    int cost_tile_h;            /* SAD / SSD kernels: output rows per wave */
    int cost_kernel;            /* 1 = the general masked kernel even where the quad-SAD / MFMA / dot kernels apply;
                                   2 = SSD on the byte dot-product unit instead of the matrix cores;
                                   3 = the ghost-border strip by the general masked kernel (as until round 3);
                                   4 = SAD by the round-4 kernel (every window row from scratch) where the
                                       prefix-chain kernel of round 5 applies */
    int priority_class;         /* bit-sliced kernel: which of a SIMD's two waves a priority slice favours is told by
                                 * 1 = the wave slot's parity, 2 = the parity of the workgroup's slot on its CU (the two
                                 * waves of a two-wave workgroup are then favoured together); 0 = the plan's choice */
    int priority_on_change;     /* bit-sliced kernel: 1 = s_setprio only when the wanted priority changes, 2 = once per row;
                                 * 0 = the plan's choice */
    int lane_merge;             /* bit-sliced kernel: how the lanes that split a word's shift range are merged:
                                 * 1 = per row with DPP, 2 = through LDS every four rows (where >= 4 lanes share a word);
                                 * 0 = the plan's choice */
    int no_four_shift_lanes;    /* bit-sliced kernel: 1 = never 4 shifts per lane (the plan's own choice is between 16, 8
                                 * and 4); shifts_per_lane = 4 forces them where they are built */
    int priority_unit_log2;     /* bit-sliced kernel: log2 of the priority schedule's unit in shader-clock cycles (8 .. 20;
                                 * bit k of priority_pattern covers the k-th unit); 0 = the plan's choice */
    int cost_workgroup_waves;   /* SAD kernel of round 5 (prefix chains): 1, 2 or 4 waves per workgroup sharing the staged
                                 * rows; 0 = the plan's choice */
} sm_plan_options;
int sm_plan_create_ex(int device, int width, int height, int num_shifts, int square_width,
                      int border, int max_pairs, const sm_plan_options *options, sm_plan **out);

/* human-readable description of the kernel variant and tiling the plan
 * selected (for logs / bench.py); the string lives as long as the plan */
const char *sm_plan_describe(const sm_plan *plan);
/* the geometry the plan selected (kernel variant, tiling, packed-image extents):
 * what analytic cost models and tests need; plain ints only */
typedef struct sm_geometry {
    int kernel;            /* 0..2 popcount kernels A/B/C, 3 generic, 4 bit-sliced */
    int window;            /* n = 2*(square_width/2)+1 */
    int shifts_per_lane;   /* shifts one lane carries */
    int shift_lanes;       /* lanes that split one pixel group's shift range */
    int threads;           /* per workgroup */
    int tile_w, tile_h;    /* output pixels per workgroup (tile_h / waves_per_workgroup rows per wave) */
    int tiles_x, tiles_y;  /* grid (x pairs in z) */
    int ext_words, ext_rows, pad_l;   /* packed edge image: u32 words per row, rows, left pad px */
    int lds_bytes;         /* dynamic LDS request per workgroup */
    int two_wave_variant;  /* bit-sliced kernel capped at two waves per SIMD */
    int edge_rows_per_wave;/* packed-image rows one wave of the edge kernel produces */
    int waves_per_workgroup; /* bit-sliced kernel: 1, or 2 = the upper and the lower half of a tile,
                              * sharing the window rows around the middle */
    int lane_merge_lds;      /* bit-sliced kernel: the lanes that split a word's shift range are merged through LDS
                              * every four rows (1) or per row with DPP (0) */
} sm_geometry;
/* sm_plan_geometry writes sizeof(sm_geometry) bytes AS THIS HEADER DECLARES IT: the struct grows at its end from
 * release to release, so a caller that may meet a newer library than the header it was compiled against (bindings,
 * plug-ins) passes the size of ITS struct to sm_plan_geometry_sized -- the fields it knows are filled in, nothing is
 * written past them; a struct newer than the library gets its unknown tail zeroed.                               */
int sm_plan_geometry(const sm_plan *plan, sm_geometry *out);
int sm_plan_geometry_sized(const sm_plan *plan, sm_geometry *out, size_t size_of_callers_struct);
/* bytes of private device workspace */
size_t sm_plan_workspace_bytes(const sm_plan *plan);
/* Narrow result maps (sm_match_wta_typed / sm_run_typed with SM_WEB_U8 / SM_WEB_U16) of the kernels that
 * have no narrow store path of their own -- every kernel but the bit-sliced one, see sm_plan_describe --
 * go through an int32 staging map of max_pairs * W * H * 4 bytes (about 265 MB for 8 pairs at 4K).  It is
 * NOT part of a new plan: a caller that only ever asks for int32 maps never pays for it.  This call
 * allocates it (counted in sm_plan_workspace_bytes from then on); without it the first narrow request does,
 * inside that call -- a hipMalloc, which synchronises the device, so callers that time their launches call
 * this next to their own allocations.  A no-op for plans that do not need the map.  SM_ERR_NOMEM on failure. */
int sm_plan_reserve_narrow(sm_plan *plan);

/* ---- step 1: edges ------------------------------------------------------ *
 * replaces find_all_edges<<<>>> (src/stereo.cu:27-92, ghost twin
 * src/stereo-ghost.cu) for `pairs` x 2 images.  Input is uint8 gray (the
 * reference's double brightness is k/256.0, src/image.c:9-15; the decision
 * is evaluated in the same IEEE double arithmetic).  Writes the packed edge
 * bits the hot path consumes into the plan workspace and, when d_edges_* are
 * non-NULL, the reference's u8 {0,1} edge images as well.                  */
int sm_find_edges(sm_plan *plan, const uint8_t *d_gray_left,
                  const uint8_t *d_gray_right, double threshold, int pairs,
                  uint8_t *d_edges_left, uint8_t *d_edges_right, void *stream);

/* Set-up for sm_find_edges that depends on the threshold only (the decision
 * tables, built and verified on the device, with one host read-back of the
 * verdict).  sm_find_edges does this by itself the first time it sees a
 * threshold; a caller that knows the threshold in advance calls this next to
 * its allocations, as the reference allocates before its timed region
 * (src/stereo.cu:296-308).                                                  */
int sm_plan_prepare_threshold(sm_plan *plan, double threshold, void *stream);

/* alternative entry to the hot path for callers that already hold u8 {0,1}
 * edge images (the arguments of fillup_matches, src/stereo.cu:127): packs
 * them into the plan workspace                                              */
int sm_load_edges(sm_plan *plan, const uint8_t *d_edges_left,
                  const uint8_t *d_edges_right, int pairs, void *stream);

/* ---- step 2: THE HOT PATH ---------------------------------------------- *
 * one fused launch that replaces fillup_matches (src/stereo.cu:127-137),
 * the NUM_SHIFTS x {cudaMemset, addup_pixels_in_square, record_score} loop
 * (src/stereo.cu:142-207) and find_highest_scoring_shifts
 * (src/stereo.cu:211-225), for the edges last given to sm_find_edges /
 * sm_load_edges.  d_web receives the winning shift 1..num_shifts per pixel
 * (int32, the reference's `web`); d_best, if non-NULL, the winning masked
 * score (the reference's `buf`, dumped as score_best-0).                    */
int sm_match_wta(sm_plan *plan, int pairs, int32_t *d_web, int32_t *d_best,
                 void *stream);

/* The same launch with a NARROW web map: the values are shift indices 1..num_shifts, so
 * they fit uint8 (num_shifts <= 255) or uint16; d_web then points to W*H elements of that
 * type per pair.  The reference's type is int32 (src/stereo.cu:304, the D2H copy of
 * src/image.cu:15-23 moves 4 bytes per pixel); a caller that takes its results to the
 * host moves 4x / 2x fewer bytes over PCIe this way.  d_best stays int32.              */
#define SM_WEB_I32 0
#define SM_WEB_U16 1
#define SM_WEB_U8  2
int sm_match_wta_typed(sm_plan *plan, int pairs, void *d_web, int web_type,
                       int32_t *d_best, void *stream);

/* Let consecutive sm_run calls overlap.  With the flag set, call i runs on one of two
 * internal streams ("lanes", i & 1), edge detection and match launch in order, into the
 * lane's own half of a double-buffered workspace; nothing orders the lanes against each
 * other, so the edge detection of call i + 1 runs beside the match kernel of call i and
 * the first waves of match i + 1 take the SIMD slots the early finishers of match i
 * leave.  `stream` waits for the call's release event: synchronising `stream` still
 * means "all results are there", and work put on `stream` after the call sees them.
 * Two consecutive calls that share anything -- overlapping result maps, a changed
 * threshold (the decision tables are rebuilt), the staging map of a narrow result from
 * a fallback kernel -- are put in order by the library; give consecutive calls their
 * own result maps to get the overlap.  At most two calls are in flight.  The caller
 * promises that, when sm_run is called, the input images are complete in memory and no
 * work of its own that is still pending (on `stream` or elsewhere) reads or writes the
 * result maps it hands over: the call is no longer ordered behind earlier work on
 * `stream`, only behind earlier sm_run calls.  enabled = 2 keeps that ordering too (one
 * more event per call, and no overlap with the call before, whose results `stream`
 * waits for): use it when the inputs are uploaded asynchronously on `stream` just
 * before sm_run.  sm_plan_kernel_ms then measures launches that share the chip with
 * their neighbours: longer each, shorter together.  Off by default.                  */
int sm_plan_set_pipelined(sm_plan *plan, int enabled);

/* Measurement aid: with capacity > 0 the plan brackets each of the next
 * `capacity` sm_match_wta / sm_run match launches with HIP events ON THE STREAM
 * THE KERNEL IS LAUNCHED ON (capacity 0 turns it off and frees the events).
 * sm_plan_kernel_ms synchronises those events and returns the mean duration
 * in milliseconds and the number of launches recorded since the last reset.  */
int sm_plan_time_kernels(sm_plan *plan, int capacity);
int sm_plan_kernel_ms(sm_plan *plan, double *mean_ms, int *launches);
/* An event record costs a few microseconds on the launch stream (measured:
 * 8 us per step with both brackets on every launch, 6 % of a 4K step), so a
 * throughput measurement brackets only every `every`-th match launch (default
 * 1 = all of them).  Resets the launch counter.                              */
int sm_plan_time_stride(sm_plan *plan, int every);

/* steps 1 + 2 back to back: uint8 pairs in, web out */
int sm_run(sm_plan *plan, const uint8_t *d_gray_left,
           const uint8_t *d_gray_right, double threshold, int pairs,
           int32_t *d_web, int32_t *d_best, void *stream);

/* sm_run_typed whose only INPUT dependency is an event -- instead of "everything enqueued on `stream` before the call"
 * (the reference synchronises on its uploads before every pair: src/stereo.cu:402-403).  `inputs_ready_event` (a
 * hipEvent_t, e.g. recorded behind the upload of this pair on a copy stream; NULL = the images are complete now) is all
 * the call waits for besides earlier calls on the same plan that share something with it (result maps, the threshold
 * tables, the narrow staging map: put in order by the library); work enqueued on `stream` AFTER the call sees the
 * results.  That freedom is what lets consecutive calls overlap, and the plan takes it by itself: a match launch of
 * fewer than 2 x 1024 waves (a lone pair up to 4K: it cannot fill the chip twice over, and the next call's edge
 * detection and first waves fit beside its tail) runs on the plan's two internal lanes as under
 * sm_plan_set_pipelined(1); larger launches, and plans of more than 128 shifts (the edge detection the overlap hides is a small
 * part of their step, and two calls sharing the chip cost more than it), run in `stream` order behind the event.  Give consecutive calls their
 * own result maps.  Inside a stream capture the event must be one recorded in the same capture.                    */
int sm_run_after(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                 double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                 void *stream, void *inputs_ready_event);

/* STREAM CAPTURE (hipStreamBeginCapture on `stream`, torch.cuda.graph): sm_run, sm_find_edges, sm_match_wta and
 * sm_cost_wta may be recorded into a graph and replayed; a plan is single-stream, so do not run it eagerly while a
 * graph that holds its launches is in flight.  What cannot be captured returns SM_ERR_ARG with a message that names
 * the remedy, and leaves the capture valid:
 *   - a threshold whose decision tables are not built yet (their set-up reads a verdict back to the host):
 *     call sm_plan_prepare_threshold before the capture begins;
 *   - a match launch with kernel timing armed (its events cannot be read back from a graph):
 *     sm_plan_time_kernels(plan, 0) before the capture;
 *   - the first narrow result of a plan whose kernel needs the int32 staging map (an allocation):
 *     sm_plan_reserve_narrow before the capture.
 * A PIPELINED plan is captured with a protocol of its own: the lane of a call leaves `stream` at an event the
 * previous captured call recorded before it joined its own lane back, so consecutive calls still run side by side
 * inside the graph, every call joins at once (a capture may end after any call), and -- unlike outside a capture --
 * every call is ordered behind the work captured on `stream` before the capture's FIRST sm_run.  (Round 4 saw a
 * crash here: its lanes waited for release events recorded before the capture began, which the runtime answers with
 * hipErrorStreamCaptureIsolation; tools/capture_probe.hip.)                                                        */

/* sm_run with a narrow web map (see sm_match_wta_typed) */
int sm_run_typed(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                 double threshold, int pairs, void *d_web, int web_type, int32_t *d_best,
                 void *stream);

/* ---- SAD / SSD cost mode: PARITY UNPINNED ---------------------------------- *
 * BASELINE.json words the hot path as "SAD/SSD cost, window aggregation,
 * arg-min"; the reference implements the edge-equality cost above and nothing
 * else, so this entry has NO reference counterpart.  Same skeleton on the uint8
 * gray images themselves: c_d = |L(x,y) - R(x+d,y)| or its square, n x n box
 * sum, best = min over d, web = 1 + the FIRST d reaching it; borders as the
 * reference treats its edge images (wrap, or zeros past the border and no taps
 * outside the image).  Defined by oracle/stereo_oracle.c smo_cost_hot_path.
 * Windows up to 25 x 25, num_shifts <= 512.  SAD windows 3 .. 21 run on the
 * quad-SAD unit, SSD windows 3 .. 11 with up to 256 shifts on the matrix cores
 * (or, sm_plan_options.cost_kernel = 2, on the byte dot-product unit), the
 * ghost border's columns x < half behind them on a kernel of their own
 * (DESIGN.md 5.4); everything else on a general kernel -- the same results by
 * definition (tests: every path against the definition).                     */
#define SM_COST_SAD 1
#define SM_COST_SSD 2
int sm_cost_wta(sm_plan *plan, const uint8_t *d_gray_left, const uint8_t *d_gray_right,
                int cost, int pairs, int32_t *d_web, int32_t *d_best, void *stream);

/* debug tap: materialise the per-shift planes the reference dumps in debug
 * builds (matches-i, score_all-i, scores-i; src/stereo.c:98-104,:158-164,
 * :189) for one shift of one pair.  Any output may be NULL.  Slow path.    */
int sm_debug_planes(sm_plan *plan, int pair, int shift, uint8_t *d_match,
                    int32_t *d_score_all, int32_t *d_scores, void *stream);

/* debug tap: the device's edge decision (the 3-vs-3 contrast test of
 * src/stereo.c:19-27) for EVERY pair of in-image side sums: d_table receives
 * 766*766 bytes, table[sa*766+sb] in {0,1}, sums in units of 1/256.  Lets a
 * test prove the device arithmetic equals the host's for a threshold.       */
int sm_debug_edge_table(int device, double threshold, uint8_t *d_table, void *stream);

/* debug tap: the same table, but decided the way sm_find_edges decides it --
 * through the per-threshold integer lo/hi tables.
 * *not_threshold_form is set to 1 if the exact test was found not to be of
 * "true prefix / false middle / true suffix" form for some left sum, in which
 * case sm_find_edges falls back to the double arithmetic.  Synchronises.     */
int sm_debug_edge_table_fast(sm_plan *plan, double threshold, uint8_t *d_table,
                             int *not_threshold_form, void *stream);

/* ---- step 3 -------------------------------------------------------------- *
 * fill_web_holes (src/stereo.cu:235-256): `times` Jacobi sweeps over pixels
 * that are 0, ping-ponging d_web and d_tmp exactly as the reference swaps
 * its pointers; *result_in_tmp tells which buffer holds the returned image.
 * Both buffers are W*H int32 per pair.                                      */
int sm_fill_web_holes(sm_plan *plan, int32_t *d_web, int32_t *d_tmp, int times,
                      int pairs, int *result_in_tmp, void *stream);

/* array_min_gpu / array_max_gpu (src/util.cu:15-45) of each pair's image;
 * d_minmax receives 2 int32 per pair: {min, max}                            */
int sm_min_max(sm_plan *plan, const int32_t *d_image, int pairs,
               int32_t *d_minmax, void *stream);

/* draw_contour_map_kernel (src/stereo.cu:261-285) with the min/max taken
 * from d_minmax (as written by sm_min_max).  A zero interval (the reference
 * divides by it) is recorded in the plan and reported by sm_plan_status.    */
int sm_draw_contour_map(sm_plan *plan, const int32_t *d_web,
                        const int32_t *d_minmax, int num_lines, int pairs,
                        uint8_t *d_out, void *stream);

/* The whole of step 3 (fill_web_holes, image min/max, draw_contour_map:
 * src/stereo.cu:325-333) with ONE synchronisation instead of one per stage.  Same
 * results as sm_fill_web_holes + sm_min_max + sm_draw_contour_map + sm_plan_status:
 * d_out receives the contour image of the hole-filled map, d_minmax its {min, max},
 * *result_in_tmp tells which of d_web / d_tmp holds the hole-filled map.  The stages
 * are queued speculating that the map has no zero pixel (hole filling then is the
 * identity; a web from sm_match_wta never has one) and one pass over the map verifies
 * it; a map that does have holes takes the staged route.  Returns SM_ERR_ZERO_DIV
 * for a zero contour interval.  Synchronises the stream.                            */
int sm_step3(sm_plan *plan, int32_t *d_web, int32_t *d_tmp, int times, int num_lines,
             int pairs, int32_t *d_minmax, uint8_t *d_out, int *result_in_tmp, void *stream);

/* synchronises the stream and returns SM_ERR_ZERO_DIV if a contour launch
 * since the last call met a zero interval, else SM_OK                       */
int sm_plan_status(sm_plan *plan, void *stream);

#ifdef __cplusplus
}
#endif
#endif
