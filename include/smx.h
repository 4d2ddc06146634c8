/*
 * smx.h -- C-ABI of the MI355X (gfx950) stereo-pair -> disparity-map library
 *          (libsmx_hip.so, built from stereo_matching_cuda_amd/csrc/).
 *
 * This is the drop-in boundary for the one hot path of hamza1030/stereo_matching_cuda:
 *   gray -> cost volume -> guided-filter aggregation + winner-take-all -> LR check -> fill.
 * Every entry point cites the reference host function it replaces (paths relative to the
 * reference's stereo_matching_cuda/ directory).  Plain pointers and sizes only.
 *
 * Two families:
 *   smx_<stage>()      host pointers in / host pointers out, synchronous -- exactly the
 *                      calling convention of the reference's per-stage wrappers, so the
 *                      reference-signature C++ functions in stereo_matching_cuda_amd/host/
 *                      (costVolume.cuh, guidedFilter.cuh, ...) are one-line forwards.
 *   smx_dev_<stage>()  device pointers, asynchronous on a caller-supplied hipStream_t
 *                      (passed as void*) of the CURRENT device, caller-supplied workspace; no
 *                      allocation, no synchronisation and no internal streams or events: a
 *                      call is a fixed sequence of kernel launches and small memsets on
 *                      `stream`, so a pair step can be captured into a hipGraph and replayed
 *                      (tests/test_gpu_parity.py::test_pair_step_is_capturable_in_a_hip_graph).
 *                      Used by the pair path, the benchmark and the multi-GPU (D-sharded)
 *                      drivers (sharded.py; include/smx_rccl.h).
 *
 * All functions return 0 on success or a negative code (SMX_E_*); smx_last_error() gives
 * the message of the last failure on the calling thread.  There is no CPU fallback: without
 * a HIP device every compute entry point fails with SMX_E_HIP.
 *
 * Layouts (reference: costVolume.cu:178, guidedFilter.cu:173,198): images row-major [y][x];
 * volumes [z][y][x] with plane stride w*h; disparity maps are float arrays holding integer
 * labels (dmin + slice index).
 */
#ifndef SMX_H
#define SMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define SMX_OK 0
#define SMX_E_ARG (-1)  /* bad argument (null pointer, non-positive size, ...) */
#define SMX_E_HIP (-2)  /* HIP runtime error or no device */
#define SMX_E_WS (-3)   /* workspace too small */

/* Tunables of the reference (SystemIncludes.h:7-24), runtime instead of macros. */
typedef struct smx_params {
    double r_w, g_w, b_w; /* R_W 0.299, G_W 0.587, B_W 0.0721 (sic)  SystemIncludes.h:7-9 */
    double alpha;         /* ALPHA 0.9                                :10 */
    int th_color;         /* TH_color 7                               :14 */
    int th_grad;          /* TH_grad 2                                :13 */
    int radius;           /* RADIUS 9                                 :21 */
    double eps;           /* EPS 6.5025                               :23 */
    int d_lr;             /* D_LR 0                                   :24 */
} smx_params;

void smx_default_params(smx_params* p);
const char* smx_last_error(void);
const char* smx_version(void);
/* Number of visible HIP devices (0 if none / runtime unusable). Never fails. */
int smx_device_count(void);

/* ------------------------------------------------------------------------------------
 * Host-pointer stage API (reference L2 wrappers)
 * ---------------------------------------------------------------------------------- */

/* rgb_to_grayscale.cuh:7  unsigned char* rgb_to_grayscale(h_rgb, n, channels, compare)
 * gray[k] = (uchar)(R_W*r + G_W*g + B_W*b) in double.  h_gray: caller-allocated, n bytes. */
int smx_rgb_to_grayscale(const smx_params* p, const uint8_t* h_rgb, int64_t n, int channels,
                         uint8_t* h_gray);

/* costVolume.cuh:7  void compute_cost(i1, i2, cost, w1, w2, h1, h2, dmin, compare)
 * size_d is explicit here (the reference derives it from macros, costVolume.cu:5).
 * cost: size_d*w1*h1 floats, [z][y][x], label of slice z = dmin + z. */
int smx_compute_cost(const smx_params* p, const uint8_t* i1, const uint8_t* i2, float* cost,
                     int w1, int w2, int h1, int h2, int size_d, int dmin);

/* integral.cuh:3  void integral(float* image, float* integral, int width, int height) */
int smx_integral(const float* image, float* integral, int width, int height);

/* guidedFilter.cuh:7  void compute_guided_filter(i, cost, filter_cost, disp_map, mean, w, h,
 *                                                size_d, dmin, compare)
 * filter_cost / disp_map are IN/OUT exactly as in the reference (main.cu:112-118 presets them
 * to 0x7F7F7F7F / 0): a pixel is updated iff filter_cost >= min_z q[z].  mean (u8, optional)
 * receives trunc(mean_I).  agg (optional, size_d*w*h floats) receives the aggregated volume q,
 * which the reference never materialises (guidedFilter.cu:233). */
int smx_compute_guided_filter(const smx_params* p, const uint8_t* i, const float* cost,
                              float* filter_cost, float* disp_map, uint8_t* mean, float* agg,
                              int w, int h, int size_d, int dmin);

/* occlusion.cuh:8  void detect_occlusion(dL, dR, dOcclusion, dmapl, dmapr, w, h)
 * (the two u8 arguments of the reference are dead: occlusion.cu:51-52). dL is in/out. */
int smx_detect_occlusion(const smx_params* p, float* disparityLeft, const float* disparityRight,
                         int dOcclusion, int w, int h);

/* occlusion.cuh:14  void fill_occlusion(float* disparity, w, h, vMin) ; in place.
 * One wave per row with the row staged in LDS: w <= 16384 (SMX_E_ARG above that). */
int smx_fill_occlusion(float* disparity, int w, int h, float vMin);

/* filter.cuh:12  void filter(image, width, height, mean, var, cuda)   (dead code in the reference:
 * never called from main.cu).  Direct (2R+1)^2 box filter with zero padding, truncated means:
 * mean = (uchar)(int)(sum(I)/(2R+1)^2); var = (float)(int)(sum(I*I)/(2R+1)^2) - mean*mean
 * (filter.cu:39-115, 143-181).  mean: w*h bytes, var: w*h floats, caller-allocated. */
int smx_filter(const smx_params* p, const uint8_t* image, int w, int h, uint8_t* mean, float* var);

/* main.cu:65-155 as one call on two gray images (device-resident between stages).
 * Left volume labels dminl .. dminl+size_d-1, right volume dminr .. dminr+size_d-1
 * (main.cu:79-82).  Any output pointer may be NULL.  occlusion uses dOcclusion = dminl-100
 * (main.cu:149) and filling uses vMin = dminl (main.cu:154). */
typedef struct smx_pair_out {
    float* best_l; float* best_r;   /* n floats each: WTA cost (main.cu best_costl/r)      */
    float* dmap_l; float* dmap_r;   /* n floats each: labels                                */
    uint8_t* mean_l; uint8_t* mean_r;
    float* occlusion;               /* left map after LR check                              */
    float* filled;                  /* after scan-line filling                              */
    float* cost_l; float* cost_r;   /* size_d*n floats each, raw cost volumes (optional)    */
    float* agg_l; float* agg_r;     /* size_d*n floats each, aggregated volumes (optional)  */
} smx_pair_out;

int smx_stereo_pair(const smx_params* p, const uint8_t* gray_l, const uint8_t* gray_r, int w,
                    int h, int size_d, int dminl, int dminr, const smx_pair_out* out);

/* Persistent context for the host-pointer pair entry: device buffers, workspace and stream are created once
 * and reused by every pair of the same shape.  It replaces the per-call allocation churn of the reference's
 * wrappers (guidedFilter.cu:50-56,182-194: 13 planes uploaded per slice; integral.cu:3-51: five cudaMalloc /
 * cudaFree per call), which smx_stereo_pair still pays once per call (it is smx_create + smx_ctx_stereo_pair +
 * smx_destroy).  The context belongs to the device that was current at smx_create; use it from one host thread
 * at a time.  cost_* / agg_* volumes are allocated on first request and kept. */
typedef struct smx_ctx smx_ctx;
int smx_create(const smx_params* p, int w, int h, int size_d, smx_ctx** ctx);
int smx_ctx_stereo_pair(smx_ctx* ctx, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr,
                        const smx_pair_out* out);
int smx_destroy(smx_ctx* ctx);
/* Pipelined host-pointer entry: uploads of pair k+1 and downloads of pair k-1 run under the aggregation of pair k
 * (three streams, pinned staging buffers owned by the context; it replaces the synchronous per-slice upload / compute /
 * download churn of guidedFilter.cu:50-56,182-194,244-248).
 *   smx_ctx_stereo_pair_async  copies the two images into pinned staging (the caller's buffers are free when it returns),
 *                              enqueues upload -> path -> download of the eight result planes and returns without
 *                              waiting.  At most two pairs may be in flight (SMX_E_ARG otherwise).
 *   smx_ctx_wait               waits for the OLDEST pair in flight.  `staged` (may be NULL) receives pointers to its
 *                              results inside the context's pinned staging -- valid until two more pairs have been
 *                              submitted; `copy_to` (may be NULL) names caller buffers the planes are copied into as
 *                              well (a host memcpy of 26 bytes per pixel: use `staged` where the rate matters).
 *                              cost_* / agg_* volumes are not part of the pipelined entry.  Returns the pair's status.
 * Results equal those of smx_ctx_stereo_pair bit for bit.  The synchronous entry refuses to run while pairs are in flight. */
int smx_ctx_stereo_pair_async(smx_ctx* ctx, const uint8_t* gray_l, const uint8_t* gray_r, int dminl, int dminr);
int smx_ctx_wait(smx_ctx* ctx, smx_pair_out* staged, const smx_pair_out* copy_to);
/* Aggregation path of this context (ids as for smx_set_agg_path below). */
int smx_ctx_set_agg_path(smx_ctx* ctx, int path);

/* ------------------------------------------------------------------------------------
 * Device-pointer API (async on `stream`, no allocation inside)
 * ---------------------------------------------------------------------------------- */

int smx_dev_rgb_to_grayscale(const smx_params* p, const uint8_t* d_rgb, int64_t n, int channels,
                             uint8_t* d_gray, void* stream);

/* Slices [s_begin, s_end) of the volume whose slice z has label dmin+z are written to
 * d_cost[(z - s_begin) * w1*h]. */
int smx_dev_cost_volume(const smx_params* p, const uint8_t* d_i1, const uint8_t* d_i2,
                        float* d_cost, int w1, int w2, int h, int dmin, int s_begin, int s_end,
                        void* stream);

/* nplanes independent w*h planes, in place allowed (d_in == d_out). */
int smx_dev_integral(const float* d_in, float* d_out, int w, int h, int nplanes, void* stream);

/* Bytes of workspace smx_dev_aggregate_wta needs for `nslices` slices of ONE view in flight (the pair
 * call needs twice that): image / guidance planes, per slice one aggregated plane plus the strip
 * hand-off records, control words.  The first 256 bytes hold the call's status word
 * (smx_dev_agg_status).  Fewer slices in flight than s_end - s_begin only means more launches. */
size_t smx_agg_workspace_bytes(int w, int h, int nslices);
/* The same for the path that `p` will run in auto mode: with radius <= 9 the fused walker needs ONE plane per slice in
 * flight (plus its hand-off records), about a quarter of the radius-agnostic bound above, which has to cover the five
 * planes per slice of the multi-kernel path (radius > 9, or smx_set_agg_path(1)).  At 3840x2160 this is what lets all 512
 * slices of a volume go out in one launch inside 64 GB. */
size_t smx_agg_workspace_bytes_for(const smx_params* p, int w, int h, int nslices);

/* Guided-filter aggregation + running winner-take-all over slices [s_begin, s_end) of ONE
 * volume (reference: guidedFilter.cu:171-238 incl. dispSelectOnGPU :403-411).
 *   d_guide  : guidance image I (u8, w*h) -- the image the volume belongs to
 *   d_other  : the other view; used to build cost slices on the fly when d_cost == NULL
 *              (costVolume.cu:163-190 fused in).  May be NULL when d_cost is given.
 *   d_cost   : optional materialised cost slices, slice s at d_cost[(s - s_begin)*w*h]
 *   d_keys   : n packed WTA keys, IN/OUT: key = sord(cost)<<32 | (0xFFFFFFFF - slice) compared as
 *              signed 64-bit integers (sord = order-preserving f32 -> i32 map); initialise with
 *              smx_dev_init_keys (INT64_MAX); combine across shards with an int64 MIN all-reduce
 *   d_mean_u8: optional u8 mean image out (guidedFilter.cu:87,122)
 *   d_agg    : optional aggregated slices out, slice s at d_agg[(s - s_begin)*w*h]; any 4-byte aligned pointer
 *              (the WTA pass reads it with 8-byte loads when it is 8-byte aligned and w*h is even, else with 4-byte ones)
 * Slices are processed in chunks that fit the workspace (>= smx_agg_workspace_bytes(w,h,1)). */
int smx_dev_aggregate_wta(const smx_params* p, const uint8_t* d_guide, const uint8_t* d_other,
                          const float* d_cost, int w, int h, int dmin, int s_begin, int s_end,
                          int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg, void* d_workspace,
                          size_t workspace_bytes, void* stream);

/* Both views of a stereo pair in one call (main.cu:133-134 back to back): every kernel launch
 * covers the left and the right volume.  View 0 = left (guide d_left, labels dminl + s), view 1 = right (guide d_right, labels
 * dminr + s).  d_keys: 2*n keys (left then right); d_mean_u8: NULL or 2*n bytes; d_agg: NULL or two
 * volumes of (s_end - s_begin)*n floats.  Workspace: >= 2 * smx_agg_workspace_bytes(w, h, nslices). */
int smx_dev_aggregate_wta_pair(const smx_params* p, const uint8_t* d_left, const uint8_t* d_right,
                               int w, int h, int dminl, int dminr, int s_begin, int s_end,
                               int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg, void* d_workspace,
                               size_t workspace_bytes, void* stream);

/* The same with the two cost volumes materialised by the caller -- main.cu:80-82 followed by main.cu:133-134, i.e. the
 * reference's own data flow: read the raw cost, write / consume the aggregated cost (guidedFilter.cu:198-233).  Slice s of a
 * volume at d_cost_*[(s - s_begin) * w*h], as for smx_dev_aggregate_wta. */
int smx_dev_aggregate_wta_pair_cost(const smx_params* p, const uint8_t* d_left, const uint8_t* d_right,
                                    const float* d_cost_l, const float* d_cost_r, int w, int h, int dminl, int dminr,
                                    int s_begin, int s_end, int64_t* d_keys, uint8_t* d_mean_u8, float* d_agg,
                                    void* d_workspace, size_t workspace_bytes, void* stream);

/* Synchronous health check of the last smx_dev_aggregate_wta[_pair] call that used d_workspace:
 * copies the call's status word back (call it after synchronising the launch stream).  SMX_E_HIP if
 * a workgroup of the fused kernel gave up waiting for another one (its left neighbour strip or the
 * guidance of its strip; the spins are bounded, 2 s), in which case the results are invalid.  The
 * waits cannot deadlock (every wait is for a work item with a smaller ticket), so this only fires
 * if the GPU is taken away mid-launch; the host-pointer wrappers call it for you. */
int smx_dev_agg_status(const void* d_workspace);
/* Materialised cost volumes (d_cost != NULL) and radius 9: the comb walker loads the costs and checks that each is +0 or a
 * normal number in [2^-60, 2^60] -- what its exactness argument covers (costVolume.cu:187 produces nothing else).  A
 * call with other values (negative, -0, denormal, infinite, NaN) is still answered bit-exactly: the ring walker is queued
 * behind the comb walker and redoes the chunk on the device when the check fired.  This reports, after synchronising the
 * stream, whether that happened in the last call that used d_workspace (the call then cost about 2.5 x). */
int smx_dev_agg_fallback(const void* d_workspace, int* ring_walker_reran);

/* Aggregation implementation of the calling THREAD's smx_dev_* and host-pointer stage calls (a persistent
 * context carries its own, smx_ctx_set_agg_path; it starts with the creating thread's):
 *   0 = auto: the fused single-kernel aggregation when radius <= 9, else the multi-kernel path; the fused call
 *       picks the comb walker (smx_agg_v5.hip: radius 9, costs built from the images) or the ring walker
 *       (smx_agg_v4.hip: any radius <= 9, materialised cost volumes)
 *   1 = force multi-kernel            2 = force fused (error if radius > 9), walker chosen as in auto
 *   3 = fused, ring walker forced     5 = fused, comb walker forced (error where it does not apply)
 *   4 = FAST, NOT bit-exact; reported separately, never a default.  Where the comb walker applies: the same sums in the
 *       same order, window means by multiplication with the rounded reciprocal of the area instead of the exact division
 *       (aggregated costs within ~2.4e-4 relative).  Elsewhere: the ring walker with wave-parallel, re-associated row
 *       prefix sums (within ~4e-3).
 * smx_last_agg_path() reports what the last aggregation on this thread ran: 1 multi-kernel, 2 ring walker,
 * 4 FAST, 5 comb walker. */
int smx_set_agg_path(int path);
int smx_last_agg_path(void);
/* Slices per launch of the fused aggregation.  The reference's slice loop (guidedFilter.cu:171-238) handles ONE slice per
 * iteration; the fused walker handles as many per launch as the caller's workspace holds, and accumulates the running WTA
 * across launches.  smx_set_max_slices_per_launch(n > 0) bounds that number for the calling thread's calls whatever the
 * workspace holds (0 = no bound, the default) -- the knob behind `slices_in_flight` of the Python pipeline and bench.py.
 * smx_last_agg_chunk reports what the calling thread's last fused aggregation did: slices per walker launch (of the first,
 * i.e. largest, launch) and the number of walker launches. */
int smx_set_max_slices_per_launch(int n);
/* d_keys of smx_dev_aggregate_wta[_pair[_cost]] is IN/OUT (shards and chunks accumulate into it; initialise with
 * smx_dev_init_keys).  smx_set_keys_fresh(1) tells the calling thread's following aggregation calls that the keys hold
 * nothing yet: the call's first WTA pass starts from the identity instead of loading them, which saves the
 * smx_dev_init_keys launch (7 us and 7.5 MB of stores per KITTI pair).  smx_set_keys_fresh(0) restores IN/OUT. */
int smx_set_keys_fresh(int on);
int smx_last_agg_chunk(int* slices_per_launch, int* walker_launches);
/* Tile geometry of the fused aggregation for a box radius: output columns per strip, rows per band,
 * columns computed per strip (strip_cols + 2*radius + 1).  For tests that aim at tile boundaries. */
int smx_agg_geometry(int radius, int* strip_cols, int* band_rows, int* tile_cols);

/* winner_take_all.cuh (live WTA = dispSelectOnGPU, guidedFilter.cu:403-411), packed form. */
int smx_dev_init_keys(int64_t* d_keys, int64_t n, void* stream);
/* Fold keys into best/dmap with the reference's rule: if (best >= q) { dmap = dmin + slice;
 * best = q; }.  best/dmap are IN/OUT (use smx_dev_init_wta for the reference's presets). */
int smx_dev_apply_keys(const int64_t* d_keys, int64_t n, int dmin, float* d_best, float* d_dmap,
                       void* stream);
/* main.cu:112-118: best <- 0x7F7F7F7F bit pattern, dmap <- 0. */
int smx_dev_init_wta(float* d_best, float* d_dmap, int64_t n, void* stream);

int smx_dev_filter(const smx_params* p, const uint8_t* d_image, int w, int h, uint8_t* d_mean,
                   float* d_var, void* stream);

int smx_dev_detect_occlusion(const smx_params* p, float* d_dL, const float* d_dR, int dOcclusion,
                             int w, int h, void* stream);
int smx_dev_fill_occlusion(float* d_disp, int w, int h, float vMin, void* stream);
/* main.cu:112-155 behind the aggregation, for both views at once: d_keys / d_best / d_dmap hold the left
 * view in [0, n) and the right one in [n, 2n), n = w*h.  best <- preset, dmap <- 0, the winning slice of
 * every key applied (smx_dev_init_wta + smx_dev_apply_keys), d_occlusion <- left map after the LR check
 * (smx_dev_detect_occlusion with dOcclusion), d_filled <- d_occlusion filled (smx_dev_fill_occlusion with
 * vMin).  One launch (three for rows of more than 8192 pixels) instead of seven; same results (tested against
 * the per-call sequence). */
int smx_dev_finish_pair(const smx_params* p, const int64_t* d_keys, int w, int h, int dminl, int dminr,
                        int dOcclusion, float vMin, float* d_best, float* d_dmap, float* d_occlusion,
                        float* d_filled, void* stream);

/* Host-side helpers for the packed key (same encoding as the kernels). */
int64_t smx_pack_key(float cost, uint32_t slice);
void smx_unpack_key(int64_t key, float* cost, uint32_t* slice);

/* Per-stage device time (ms) of the calling thread's most recent timed call -- smx_dev_aggregate_wta[_pair]
 * (+ a following smx_dev_finish_pair) or smx_ctx_stereo_pair -- when timing was enabled with smx_set_timing(1):
 * HIP events of the device the call ran on, recorded on its stream at the stage boundaries (the reference prints
 * one wall-clock `duration`, main.cu:52-54,156,184).  smx_stage_times synchronises with the last event.
 *   upload / download: host <-> device copies of smx_ctx_stereo_pair;  guidance: key presets, image planes,
 *   guidance statistics (guidedFilter.cu:58-123);  aggregation: the fused walker (or the multi-kernel passes);
 *   wta: the packed-key pass over the aggregated planes;  finish: decode, LR check, filling (main.cu:112-155).
 * smx_set_timing: 0 off, 1 the times of the LAST call, 2 cumulative over every call since it was switched on
 * (`calls` counts them; up to 32768 stage marks, later ones are counted in `dropped`).  The pipelined entry
 * (smx_ctx_stereo_pair_async) records no stage marks. */
typedef struct smx_stage_ms {
    float upload, guidance, aggregation, wta, finish, download, total;
    int calls;
    int dropped;   /* stage marks that did not fit (mode 2 keeps 32768): > 0 means the sums cover only the first calls */
} smx_stage_ms;
int smx_set_timing(int mode);
int smx_stage_times(smx_stage_ms* out);
/* guidance + aggregation + wta of that call, and its kernel launches */
int smx_last_agg_ms(float* ms, int* launches);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SMX_H */
