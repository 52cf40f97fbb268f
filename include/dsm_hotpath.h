/*
 * dsm_hotpath.h -- C ABI of the MI355X-native direct photometric hot path.
 *
 * Drop-in boundary for the hot path of IRVLab/direct_stereo_slam.
 * The reference has no FFI layer: the seam is the C++ class `dso::TrackerAndScaler`
 * (src/scale_optimization/TrackerAndScaler.h:34-137) and the free functions
 * `search_ringkey` / `search_sc` (src/loop_closure/loop_detection/search_place.h:25-84).
 * Every entry point below names the reference interface it replaces (file:line,
 * relative to the reference tree).  The host adaptor that keeps the reference's C++
 * surface on top of this ABI is direct_stereo_slam_amd/host/TrackerAndScaler.hpp; the
 * reference-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, POD only, no exceptions; every function returns DSM_OK (0) or a negative
 *    dsm_status.  dsm_last_error() gives a thread-local human readable message.
 *  - all image / template data is IEEE float32; poses and affine brightness are float64
 *    exactly as the reference keeps them (Sophus SE3d, AffLight{double a,b}).
 *  - pose layout: double[7] = {qx, qy, qz, qw, tx, ty, tz} (Eigen quaternion coefficient
 *    order, then translation) for the transform refToNew / lastToNew.
 *  - host pointers are read during the call and never retained after it returns.
 *  - one HIP stream per context; one call in flight per context (the reference calls
 *    this path under track_mutex_ / coarse_tracker_swap_mutex_, FrontEnd.cpp:589,628).
 *  - there is NO CPU fallback: every entry point fails with DSM_ERR_NO_DEVICE when no
 *    gfx950 device is usable.
 */
#ifndef DSM_HOTPATH_H
#define DSM_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_ABI_VERSION 4 /* 2: dsm_params.struct_size / fixed_schedule / frame_check / frame_grad_tol, dsm_stats.evals_residual_only;
                             3: dsm_params_default_sized / DSM_PARAMS_INIT, dsm_stream_* (streaming form of the batched calls);
                             4: dsm_params.chunk_geometry takes the slot of version 3's tile_l0 -- same layout, another meaning: 0 now
                                selects the THROUGHPUT chunk table (float sums differ in their last bits from versions <= 3, whose only
                                table is today's 1 = LATENCY), and a version-3 caller's tile_l0 = 1 would silently select the latency
                                table -- hence the bump; dsm_set_refs_from_points requires one geometry per call */
#define DSM_MAX_LEVELS 6 /* DSO PYR_LEVELS; the reference tracker uses <= 5 (TrackerAndScaler.cpp:457,463) */

typedef enum dsm_status {
  DSM_OK = 0,
  DSM_ERR_INVALID = -1,   /* bad argument */
  DSM_ERR_NO_DEVICE = -2, /* no usable HIP device */
  DSM_ERR_HIP = -3,       /* HIP runtime error, see dsm_last_error() */
  DSM_ERR_STATE = -4,     /* call order violated (e.g. track before set_ref) */
  DSM_ERR_NOMEM = -5
} dsm_status;

typedef struct dsm_context dsm_context; /* device + stream + workspaces */
typedef struct dsm_tracker dsm_tracker; /* one TrackerAndScaler instance */
typedef struct dsm_ringdb dsm_ringdb;   /* ring-key database + delay queue */
typedef struct dsm_comm dsm_comm;       /* communicator of the sharded ring-key database: one rank per GPU (RCCL) */
typedef struct dsm_pose_estimator dsm_pose_estimator; /* loop-closure direct alignment (PoseEstimator) */

/* Runtime parameters.  These are DSO globals / literals in the reference; the values
 * written by dsm_params_default() are the upstream DSO defaults as used by the
 * reference's default `mode=1` (src/main.cpp:117-121). */
typedef struct dsm_params {
  size_t struct_size;                 /* sizeof(dsm_params) of the header the CALLER was built with: set by DSM_PARAMS_INIT /
                                         dsm_params_default_sized from the size the caller's compiler sees; every entry point
                                         that takes parameters returns DSM_ERR_INVALID on a mismatch, so a host built against
                                         another version of this header fails loudly instead of handing over a shorter
                                         struct.  Hosts should also compare dsm_abi_version() with DSM_ABI_VERSION once (the
                                         Python and C++ adaptors in this repository do). */
  float huber_th;                     /* setting_huberTH            (TrackerAndScaler.cpp:727,795)   9    */
  float coarse_cutoff_th;             /* setting_coarseCutoffTH     (TrackerAndScaler.cpp:476)       20   */
  float scale_xi_rot;                 /* SCALE_XI_ROT               (TrackerAndScaler.cpp:542,685)   1    */
  float scale_xi_trans;               /* SCALE_XI_TRANS             (TrackerAndScaler.cpp:543,686)   0.5  */
  float scale_a;                      /* SCALE_A                    (TrackerAndScaler.cpp:544,687)   10   */
  float scale_b;                      /* SCALE_B                    (TrackerAndScaler.cpp:545,688)   1000 */
  float affine_opt_mode_a;            /* setting_affineOptModeA     (TrackerAndScaler.cpp:511-534)   0    */
  float affine_opt_mode_b;            /* setting_affineOptModeB                                      0    */
  float lambda_extrapolation_limit;   /* literal                    (TrackerAndScaler.cpp:464,863)   0.001*/
  int max_iterations[DSM_MAX_LEVELS]; /* literal {10,20,50,50,50}   (TrackerAndScaler.cpp:463,862); [5]=50 is an extension */
  int adaptive_schedule;              /* 1 (default): speculative per-level launch counts learnt from previous calls, one
                                         host read-back per pass (plus, with compact_tail, one per level and stream group in the
                                         first pass); 0: enqueue the worst case (2*(7+max_iterations) launch
                                         pairs per level) and never poll.  Scheduling only -- results are identical. */
  int persistent_coarse;              /* N > 0: pyramid levels whose target plane has at most min(N, 9216) pixels (156x48,
                                         120x67, ...; and at most 20 chunks of template points) run their whole LM loop inside
                                         ONE kernel launch per problem, on an LDS-resident copy of the plane (and of the
                                         template when both fit), speculative candidates included; 0 (default): one
                                         (evaluate, step) launch pair per LM evaluation at every level.  Scheduling only --
                                         results are bit-identical.  Measured: neutral for hundreds of frames in
                                         flight (a third of the launches), slower for one frame (the launch form spreads a
                                         level's chunks over many CUs).
                                         N < 0 (round 6): the CHAIN form for single calls -- every level whose evaluation is ONE chunk
                                         (chunk_geometry) runs its LM loop in one launch per problem (evaluate, reduce, step on an LDS
                                         copy of the state; global gathers, no speculative candidates), down to the first level of
                                         several chunks, where the launch-per-step schedule takes over: one frame in flight saves a
                                         launch per LM round of its coarse levels (with chunk_geometry 2: what the replay adaptors
                                         set).  Scheduling only -- bit-identical under the same chunk table. */
  int fuse_lm;                        /* at pyramid levels >= 1 the evaluation kernel's last-arriving workgroup of a
                                         problem can perform the LM step itself (one launch per evaluation instead of
                                         two): 0 never, 1 (default) for batches of at most 8 problems (where it shortens
                                         the latency chain), 2 always.  Scheduling only -- results are bit-identical. */
  int work_queue;                     /* the whole call as ONE launch of persistent workgroups that pull (problem, chunk)
                                         items from a device-side queue; the workgroup completing an evaluation performs
                                         the LM step and enqueues the problem's next evaluation, so problems advance
                                         independently instead of in lock-step launches: 0 never, 1 (default) for batches
                                         of at least 32 problems with at most 24576 finest-level chunks in all (beyond
                                         that the lock-step launches are faster), 2 always.  Scheduling only -- results
                                         are bit-identical. */
  int speculate;                      /* launch-per-step form: next to the LM proposal being evaluated, the proposal that
                                         FOLLOWS IF IT IS REJECTED (same H, b, pose; lambda x 4, TrackerAndScaler.cpp:583-585)
                                         is evaluated in the same launch, so a rejected step -- about half of all steps on
                                         the coarse levels -- costs no launch of its own.  Same evaluations, decisions and
                                         per-level evaluation counts as the sequential loop; the speculative evaluation of
                                         an ACCEPTED step is discarded (not counted).  0 never, 1 (default) on levels of at most
                                         8192 template points whose launches evaluate at most a million points per stream
                                         group (where launches are latency-bound and rejections come in runs),
                                         2 on every level.  Scheduling only -- results are bit-identical. */
  int compact_tail;                   /* launch-per-step form, batches: 1 (default) after the rounds most problems of a level
                                         need (learnt from previous calls) the host reads the level's status back ONCE and
                                         runs the remaining rounds -- needed by a few stragglers -- as compact launches over
                                         those problems only, and every later pass over the problems still running; 0 every
                                         launch covers every problem (idle workgroups cost ~1.6 ns each: 100 us for a level-0
                                         launch over 512 problems).  Scheduling only -- results are bit-identical. */
  int fixed_schedule;                 /* K > 0: BENCHMARK schedule of SURVEY.md section 8d -- every level runs exactly one
                                         initial evaluation + K LM iterations whose steps are ALL taken (no accept test, no
                                         cut-off repeat, no small-step break, no abort), so that the evaluations and bytes
                                         per frame do not depend on the input.  Not the reference's algorithm: 0 (default)
                                         runs trackNewestCoarse / optimizeScale as written. */
  int chunk_geometry;                 /* Reduction geometry: which table gives a chunk's points per thread from the level's point count
                                         (a chunk = one workgroup's share of an evaluation = one partial).
                                         0 (default) THROUGHPUT: 16 points per thread; a level of at most 4096 points is ONE chunk (the smallest
                                           of 1 / 2 / 4 / 8 / 16 points per thread that holds it) --
                                           long chunks, their fixed cost (three dependent memory round trips before the first
                                           point, the 52-sum reduction after the last) amortised: many problems in flight.
                                           (ABI version 3's table 0 -- 16 from 16 k points up, 8 / 4 / 2 from 4 k / 1 k / 512 -- gave the
                                           small levels several short chunks; version 4's float sums there differ in their last bits.)
                                         1 LATENCY: 16 / 8 / 4 / 2 from 256 k / 64 k / 16 k / 4 k points, else 1 -- short chunks, more
                                           workgroups per evaluation, a single evaluation through sooner: ONE problem in flight
                                           Rounds 1-4 used this table for everything.
                                         2 CHAIN (round 6): table 1 above 4096 points, ONE chunk up to 4096 points (as table 0): one problem
                                           in flight whose small levels run as a chain (persistent_coarse < 0; the tick engine's chains):
                                           what the replay adaptors set.
                                         Every form (single, batch, stream) follows the tracker's table: results of one tracker are
                                         bit-identical across forms; between the tables only the summation tree of the float sums
                                         differs (last bits), integer outputs are equal.  dsm_reduction_geometry reports the choice.
                                         (This slot was round 4's tile_l0, RESERVED = 0 since: a caller that leaves it 0 gets the default.) */
  int frame_check;                    /* dsm_tracker_upload_frame: 1 (default) verify that the caller's gradient channels
                                         are makeImages' central differences of channel 0 (the device stores channel 0 only);
                                         0 trust the caller (channels 1, 2 are ignored) */
  float frame_grad_tol;               /* with frame_check = 1: 0 (default) bitwise equality; t > 0 accepts
                                         |g - g'| <= t * max(1, |g'|) per texel and channel (a host whose makeImages was built
                                         with other floating-point flags) */
} dsm_params;

/* Statistics of the last track / optimize_scale (batch) call on a context. */
typedef struct dsm_stats {
  int64_t evals[DSM_MAX_LEVELS];        /* evaluations executed, summed over the batch: fused residual + Jacobian (calcRes* +
                                           calcGSSSE*), except the evals_residual_only below */
  int64_t launches[DSM_MAX_LEVELS];     /* eval kernel launches per level */
  int64_t algorithmic_bytes;            /* sum over evals of 16*n_l + min(12*w_l*h_l, 48*n_l)   */
  double eval_kernel_ms[DSM_MAX_LEVELS];/* summed HIP-event durations of the eval kernel dispatches per level (timing enabled) */
  double eval_kernel_union_ms[DSM_MAX_LEVELS]; /* time during which at least one of them ran (stream groups overlap) */
  int64_t eval_dispatches[DSM_MAX_LEVELS];     /* timed dispatches per level (launches x stream groups) */
  double total_ms;                      /* HIP-event time of the whole call */
  int64_t polls;                        /* host read-backs of the device LM state (passes) */
  int64_t coarse_launches;              /* launches of the persistent small-level kernel */
  int64_t queue_blocks;                 /* work-queue kernel: persistent workgroups launched (0: launch-per-step form) */
  int64_t queue_items;                  /* work-queue kernel: (problem, chunk) items processed */
  double queue_kernel_ms;               /* work-queue kernel: HIP-event duration of the launch (timing enabled) */
  int64_t evals_residual_only[DSM_MAX_LEVELS]; /* of evals[]: the last evaluation of a level's LM loop when the loop is known
                                           to end after it (increment below 1e-3, TrackerAndScaler.cpp:588 / :937, or the
                                           iteration bound): the reference runs calcGSSSE* on it and never reads the result;
                                           here only calcRes* is executed for it */
} dsm_stats;

const char *dsm_last_error(void);
int dsm_abi_version(void);
/* Defaults into the CALLER's struct: caller_size = sizeof(dsm_params) as the caller's compiler sees it.  The library
 * writes min(caller_size, its own sizeof) bytes -- never past the end of the caller's struct -- records caller_size in
 * struct_size, and returns DSM_ERR_INVALID when the two sizes differ (the caller was built against another header; every
 * entry point taking the struct would refuse it too).  Use the macro. */
int dsm_params_default_sized(dsm_params *p, size_t caller_size);
#define DSM_PARAMS_INIT(p) dsm_params_default_sized((p), sizeof(dsm_params))
/* the same without the size: ONLY for callers known to be built with this very header (the Python binding checks
 * dsm_abi_version() first): it writes the library's sizeof(dsm_params) bytes */
void dsm_params_default(dsm_params *p);

/* ---- context ------------------------------------------------------------------------- */
int dsm_context_create(int device_ordinal, dsm_context **out);
int dsm_context_destroy(dsm_context *ctx);
int dsm_context_sync(dsm_context *ctx);
/* enable per-level HIP-event timing of the eval kernels (off by default) */
int dsm_context_set_timing(dsm_context *ctx, int enable);
int dsm_context_get_stats(dsm_context *ctx, dsm_stats *out);
/* batched calls: split the batch into n_streams groups on separate HIP streams so that the small
 * (latency-bound) kernels of one group overlap the kernels of another; results are unchanged.  Default 1. */
int dsm_context_set_streams(dsm_context *ctx, int n_streams);
/* raw hipStream_t of the context (for callers that order their own device work) */
void *dsm_context_stream(dsm_context *ctx);
/* The stream groups (dsm_context_set_streams, dsm_stream_*) only overlap if their HIP streams sit on different HARDWARE queues; the
 * runtime hands its few queues (GPU_MAX_HW_QUEUES, 4 by default) out round robin to every stream of the process.  The library therefore
 * probes each stream it creates for a group (a resident kernel on one, an empty kernel on the other: docs in dsm_capi.hip) and keeps
 * only streams that run concurrently with the groups it already has.  streams_out: group streams in use (the context's own included),
 * sharing_out: how many of them had to share a queue with another group after 12 candidates (0 on a default runtime). */
int dsm_context_stream_queues(dsm_context *ctx, int *streams_out, int *sharing_out);

/* measurement aid (no reference counterpart): read-only streaming bandwidth of the device in GB/s,
 * `bytes` per pass (choose > 256 MiB to defeat the Infinity Cache), `iters` timed passes */
int dsm_diag_read_bandwidth(dsm_context *ctx, size_t bytes, int iters, double *gbps_out);
/* same, but each workgroup streams its own contiguous chunk of `chunk_bytes` (the access pattern of the
 * evaluation kernels) instead of all workgroups advancing through memory side by side */
int dsm_diag_read_bandwidth_chunked(dsm_context *ctx, size_t bytes, size_t chunk_bytes, int iters, double *gbps_out);
/* test aid: message-passing litmus of the hand-off protocol the kernels use between workgroups (device-scope stores,
 * drained store queue, ticket; device-scope loads on the other side): `pairs` producer / consumer workgroup pairs on
 * different XCDs exchange `iters` 256-byte blocks each under uneven load; *stale_words_out = words read that were not the
 * announced hand-off's (must be 0). */
int dsm_diag_xwg_litmus(dsm_context *ctx, int pairs, int iters, long long *handoffs_out, long long *stale_words_out);

/* ---- TrackerAndScaler ---------------------------------------------------------------- */
/* replaces TrackerAndScaler::TrackerAndScaler(w,h,tfm_vec,K1)  (TrackerAndScaler.cpp:47-109).
 * T_f1_f0: row-major 4x4 stereo extrinsic (cams/<set>/T_stereo.yaml); K1 = {fx,fy,cx,cy} of camera 1. */
int dsm_tracker_create(dsm_context *ctx, int w, int h, int nlevels, const double T_f1_f0[16],
                       const float K1[4], const dsm_params *params, dsm_tracker **out);
/* replaces ~TrackerAndScaler (TrackerAndScaler.cpp:111-115) */
int dsm_tracker_destroy(dsm_tracker *t);
/* replaces TrackerAndScaler::makeK(CalibHessian*) (TrackerAndScaler.cpp:117-141) */
int dsm_tracker_make_k(dsm_tracker *t, float fx, float fy, float cx, float cy);
/* replaces the OUTPUT of TrackerAndScaler::setCoarseTrackingRef (TrackerAndScaler.cpp:317-327):
 * the per-level template lists pc_u/pc_v/pc_idepth/pc_color with pc_n (built by
 * makeCoarseDepthL0 :143-315, or by dsm_make_coarse_depth_l0 below), plus refFrameID,
 * lastRef_aff_g2l and lastRef->ab_exposure. */
int dsm_tracker_set_ref(dsm_tracker *t, int ref_frame_id, double ref_aff_a, double ref_aff_b,
                        float ref_exposure, const int *n, const float *const *pc_u,
                        const float *const *pc_v, const float *const *pc_idepth,
                        const float *const *pc_color);
/* "next" row N3: makeCoarseDepthL0 + setCoarseTrackingRef (TrackerAndScaler.cpp:143-327) on the device.
 * The window's active points as flat arrays -- (pu,pv) = centerProjectedTo[0..1], pidepth =
 * centerProjectedTo[2], pweight = sqrtf(1e-3/(HdiF+1e-12)) (:155-158) -- are splatted, pyramided, dilated
 * and emitted (row-major, the reference's order) straight into this tracker's template; the keyframe's
 * (I,dx,dy) pyramid is the one already resident in `frame_owner`'s `slot` (the tracker that tracked the frame
 * which became the keyframe; may be `t` itself).  Result identical to dsm_make_coarse_depth_l0 +
 * dsm_tracker_set_ref, without the host loops and the template upload.  n_out (may be NULL): pc_n per level. */
int dsm_tracker_set_ref_from_points(dsm_tracker *t, dsm_tracker *frame_owner, int slot, int ref_frame_id,
                                    double ref_aff_a, double ref_aff_b, float ref_exposure, int npts,
                                    const float *pu, const float *pv, const float *pidepth, const float *pweight,
                                    int *n_out);
/* The same for the new keyframes of several sequences in ONE call (a node serving many sequences, dsm_stream_*): every job's
 * splat / pyramid / dilate / emit launches are enqueued back to back and the host waits ONCE for all the per-level counts.
 * A tracker may appear as `t` in one job only; results per job as dsm_tracker_set_ref_from_points.  On an error no tracker of
 * the batch has a valid reference any more.
 * ONE GEOMETRY PER CALL: every job's tracker must share image width, height and level count (the launches are batched over the
 * jobs: blockIdx.y = job); a call that mixes geometries returns DSM_ERR_INVALID before anything is enqueued -- callers with
 * trackers of several sizes issue one call per size (the per-job loop of ABI versions <= 3 accepted mixed sizes). */
typedef struct dsm_ref_job {
  dsm_tracker *t, *frame_owner;
  int slot, ref_frame_id;
  double ref_aff_a, ref_aff_b;
  float ref_exposure;
  int npts;
  const float *pu, *pv, *pidepth, *pweight;
  int *n_out; /* may be NULL */
} dsm_ref_job;
int dsm_set_refs_from_points(dsm_context *ctx, int n_jobs, const dsm_ref_job *jobs);
/* replaces TrackerAndScaler::scaleCoarseDepthL0(scale) (TrackerAndScaler.cpp:329-336) */
int dsm_tracker_scale_depth(dsm_tracker *t, float scale);
/* read back the device template of one level (tests; debugPlotIDepthMap replacement) */
int dsm_tracker_get_template(dsm_tracker *t, int lvl, int *n, float *pc_u, float *pc_v,
                             float *pc_idepth, float *pc_color);

enum { DSM_SLOT_NEW_LEFT = 0, DSM_SLOT_NEW_RIGHT = 1 };
/* replaces the consumption of FrameHessian::dIp[lvl] (TrackerAndScaler.cpp:709,1016):
 * dIp[lvl] is the reference's AoS Eigen::Vector3f (I,dx,dy) array of w_l*h_l texels.  The device keeps channel 0 only
 * and forms the gradients where they are interpolated, from the neighbouring intensities, exactly as
 * FrameHessian::makeImages (upstream DSO) defines them -- dx = 0.5 (I[idx+1] - I[idx-1]), dy = 0.5 (I[idx+w] - I[idx-w]),
 * zero where not finite -- which is the only way the reference ever fills channels 1 and 2.  The call verifies that
 * (rows 1 .. h_l-2; makeImages leaves the first and last row unset and the tracker never reads them; bitwise, or to
 * dsm_params.frame_grad_tol) and returns DSM_ERR_INVALID -- naming the first offending level and texel -- for texels that
 * were built any other way; dsm_params.frame_check = 0 skips the check. */
int dsm_tracker_upload_frame(dsm_tracker *t, int slot, const float *const *dIp, float ab_exposure);
/* the same hand-over without the gradient channels: I[lvl] = the w_l*h_l intensities of level lvl (channel 0 of dIp[lvl]),
 * 4 bytes per texel over the link instead of 12 and nothing to verify */
int dsm_tracker_upload_intensity(dsm_tracker *t, int slot, const float *const *I, float ab_exposure);
/* "next" row N1: build the pyramid on the device from the level-0 float image (upstream DSO FrameHessian::makeImages,
 * call sites FrontEnd.cpp:605,680): the intensity plane of every level; see dsm_tracker_upload_frame. */
int dsm_tracker_upload_image(dsm_tracker *t, int slot, const float *image, float ab_exposure);
/* the same hand-over for many frames in ONE call (the frames of a batched track / scale call): host->device copies
 * back to back on a copy stream, pyramids of a group of images built under the copies of the next group, five batched
 * launches per group instead of five per image.  images[i]: level-0 pixels of trackers[i]'s geometry (all trackers of one
 * call share w, h, levels), DSM_PIXEL_F32 (the undistorted float image FrontEnd.cpp:605,680 consume) or DSM_PIXEL_U8
 * (camera bytes, main.cpp:216-217 "mono8"; converted exactly on the device -- valid when no photometric calibration is
 * applied, i.e. float(pixel) is what the reference's undistorter hands on).  row_pitch_bytes: distance between image
 * rows in the caller's buffers, 0 = tight; with a pointer to the crop origin this applies the calibration file's crop
 * (cams/kitti/0_2/camera0.txt:2-4).  Returns when every copy has completed. */
enum { DSM_PIXEL_F32 = 0, DSM_PIXEL_U8 = 1 };
int dsm_upload_images(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                      const float *ab_exposures, int pixel_type, size_t row_pitch_bytes);
/* Double-buffered hand-over, so that the images of step i+1 travel while step i is being tracked:
 *   slots DSM_SLOT_NEXT_LEFT / DSM_SLOT_NEXT_RIGHT name the BACK buffers of the two frame slots (allocated on first
 *   use); dsm_frames_advance swaps back and front of the given (tracker, slot in {0,1}) pairs -- a host-side pointer
 *   swap -- and orders the context's stream after the pyramids of the last asynchronous hand-over.
 *   dsm_upload_images_async enqueues copies and pyramid kernels on a stream of its own and returns at once; the caller's
 *   buffers must stay untouched until dsm_upload_wait returns (pinned buffers are read by the GPU itself).  Track /
 *   scale calls issued meanwhile read the front buffers only.  No reference counterpart: the reference processes one
 *   frame at a time on one thread (FrontEnd.cpp:589). */
enum { DSM_SLOT_NEXT_LEFT = 2, DSM_SLOT_NEXT_RIGHT = 3 };
int dsm_upload_images_async(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                            const float *ab_exposures, int pixel_type, size_t row_pitch_bytes);
int dsm_upload_wait(dsm_context *ctx);
/* dsm_upload_images into the FRONT buffers, stream-ordered: copies and pyramid kernels are enqueued on the context's stream -- every
 * track / scale call and every dsm_stream_advance issued afterwards finds the frames in place -- and the call returns WITHOUT waiting
 * for the copies.  For page-locked capture buffers that outlive the hand-over (a node's frame ring): the caller's buffers must stay
 * untouched until dsm_upload_wait, the next dsm_upload_images* call on the context or any call that synchronises its stream.
 * Pageable buffers are handed over as dsm_upload_images does (the call then waits for their copies).  No reference counterpart. */
int dsm_upload_images_enqueue(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                              const float *ab_exposures, int pixel_type, size_t row_pitch_bytes);
int dsm_frames_advance(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots);
/* pinned host memory for images handed to dsm_tracker_upload_image (straight DMA instead of a staged copy); no reference
 * counterpart -- the reference keeps its images in ordinary host memory */
int dsm_host_alloc(size_t bytes, void **out);
int dsm_host_free(void *p);
/* read back one pyramid level as the reference's AoS (I,dx,dy) texels (gradients as makeImages forms them, zero in the
 * first and last row) */
int dsm_tracker_get_frame(dsm_tracker *t, int slot, int lvl, float *dIp_out);

/* replaces calcResPose + calcGSSSEPose as ONE fused evaluation (TrackerAndScaler.cpp:699-852, 640-697).
 * rs[6] as the reference's Vec6; H row-major 8x8 and b[8] in double, already SCALE_*-scaled.
 * n_warped = pose_buf_warped_n_ (padded to a multiple of 4). */
int dsm_tracker_calc_res_pose(dsm_tracker *t, int lvl, const double pose[7], const double aff[2],
                              float cutoff_th, double rs[6], double H[64], double b[8],
                              int *n_warped);
/* replaces calcResScale + calcGSSSEScale (TrackerAndScaler.cpp:1007-1172, 966-1005) */
int dsm_tracker_calc_res_scale(dsm_tracker *t, int lvl, float scale, float cutoff_th, double rs[6],
                               float *H, float *b, int *n_warped);

/* replaces TrackerAndScaler::trackNewestCoarse (TrackerAndScaler.cpp:451-638).
 * min_res_for_abort / last_residuals: DSM_MAX_LEVELS doubles (reference: Vec5; NaN = no limit).
 * flow_out = lastFlowIndicators.  *good = the reference's bool return value. */
int dsm_tracker_track(dsm_tracker *t, double pose_io[7], double aff_io[2], int coarsest_lvl,
                      const double *min_res_for_abort, double *last_residuals, double flow_out[3],
                      int *good);
/* replaces TrackerAndScaler::optimizeScale (TrackerAndScaler.cpp:854-964); err_out = return value */
int dsm_tracker_optimize_scale(dsm_tracker *t, float *scale_io, int coarsest_lvl, float *err_out);
/* replaces the untrapped branch of FrontEnd::optimizeScale (FrontEnd.cpp:995-1003): optimizeScale from each of n_guesses
 * initial scales (the reference's list: 0.1, 1, 5, 10, 15, 25, 30, 50) as ONE batched call on the tracker's template and right
 * frame; *scale_out / *err_out = the result with the smallest positive error (the first one on ties; 1.0 / -1 when none is
 * positive), scales_all / errs_all (may be NULL) = every guess's result in order. */
int dsm_tracker_optimize_scale_guesses(dsm_tracker *t, int n_guesses, const float *scale_guesses, int coarsest_lvl,
                                       float *scale_out, float *err_out, float *scales_all, float *errs_all);
/* One step of many sequences in ONE call: trackNewestCoarse for the n trackers of `trackers` (arguments as dsm_track_batch) and
 * optimizeScale for the n_scale trackers of `scale_trackers` (arguments as dsm_optimize_scale_batch; a tracker may appear in
 * both lists -- a keyframe's tracker is tracked against AND scale-optimised, FrontEnd.cpp:204-206,992-998).  The two sets are
 * independent problems; the scale problems' launches run on a stream of their own under the tracking kernels.  Results are
 * bit-identical to the two separate calls.  Statistics: dsm_context_get_stats (tracking) / dsm_context_get_stats2 (scale). */
int dsm_track_and_scale_batch(dsm_context *ctx, int n, dsm_tracker *const *trackers, double *pose_io, double *aff_io, int coarsest_lvl,
                              const double *min_res_for_abort, double *last_residuals, double *flow_out, int *good, int n_scale,
                              dsm_tracker *const *scale_trackers, float *scale_io, float *err_out);
int dsm_context_get_stats2(dsm_context *ctx, dsm_stats *out);

/* ---- streaming form of the batched calls: continuous admission ------------------------------
 * dsm_track_and_scale_batch takes every problem of a batch from its first evaluation to its last before it returns, so
 * its lock-step launches run as many rounds per level as the SLOWEST problem needs.  The per-frame chain
 * (TrackerAndScaler.cpp:505-593) is sequential, frames of different sequences are independent (FrontEnd.cpp:585-686):
 * nothing requires them to start and end together.  A dsm_stream is a pool of resident problems (slots) advanced in
 * PASSES: one pass sweeps the pyramid once, coarsest level first, with the rounds per level that MOST problems need (a
 * quantile of what recently retired problems took); a problem that needs more rounds at a level simply stays there
 * (its state is device-resident and resumable) and rides with the next pass's launches at that level; problems retire
 * individually and their slots are refilled from the waiting queue.  One state read-back per pass, none inside it.
 * Results are bit-identical to the batch calls (same evaluations, steps and summation order per problem).
 *   submit_* queue problems (host copies are taken; the trackers' templates and frames must stay untouched until the
 *   problem's result has been returned; the same tracker may be resident several times -- it is only read);
 *   advance  admits waiting problems into free slots, runs one pass, retires what finished (synchronous);
 *   results  hands back retired problems in retirement order;  drain = advance until nothing is resident or waiting. */
typedef struct dsm_stream dsm_stream;
typedef struct dsm_stream_result {
  uint64_t ticket;   /* from dsm_stream_submit_* */
  int kind;          /* 0 trackNewestCoarse, 1 optimizeScale */
  int good;          /* track: the reference's bool return value; scale: 1 */
  int status;        /* raw end state: 2 good, 3 aborted (:598), 4 implausible affine parameters (:615-626) */
  int passes;        /* passes the problem was resident for */
  double pose[7];    /* track: lastToNew_out as dsm_track_batch leaves pose_io (the initial guess after an abort) */
  double aff[2];     /* track: aff_g2l_out, likewise */
  double last_residuals[DSM_MAX_LEVELS];
  double flow[3];    /* track: lastFlowIndicators */
  float scale, err;  /* scale: optimizeScale's scale (:954) and return value (:963) */
  int64_t evals[DSM_MAX_LEVELS]; /* evaluations executed per level (equal to the batch form's) */
} dsm_stream_result;
int dsm_stream_create(dsm_context *ctx, int track_slots, int scale_slots, dsm_stream **out);
int dsm_stream_destroy(dsm_stream *s);
/* arguments as dsm_track_batch / dsm_optimize_scale_batch; tickets_out (may be NULL): n tickets */
int dsm_stream_submit_track(dsm_stream *s, int n, dsm_tracker *const *trackers, const double *pose0, const double *aff0,
                            int coarsest_lvl, const double *min_res_for_abort, uint64_t *tickets_out);
int dsm_stream_submit_scale(dsm_stream *s, int n, dsm_tracker *const *trackers, const float *scale0, int coarsest_lvl,
                            uint64_t *tickets_out);
/* One advance.  Tick engine: PIPELINED by default -- the call returns once the advance is enqueued and reads back the PREVIOUS one
 * (whose kernels ran while the host prepared this one), so results surface one advance late and the device never waits for the
 * host; dsm_stream_sync waits for everything in flight and reads it back (dsm_stream_drain does so itself);
 * dsm_stream_set_pipelined(s, 0) makes every advance synchronous.  Pass engine: synchronous. */
int dsm_stream_advance(dsm_stream *s);
int dsm_stream_sync(dsm_stream *s);
int dsm_stream_set_pipelined(dsm_stream *s, int on);
int dsm_stream_drain(dsm_stream *s);
int dsm_stream_results(dsm_stream *s, int max_results, dsm_stream_result *out, int *n_out);
/* as of the last read-back: problems in a slot / waiting (on the host or in the device's waiting ring) / results ready */
int dsm_stream_counts(dsm_stream *s, int *resident_out, int *waiting_out, int *results_out);
/* The engine behind advance (set before the first advance / while nothing is resident):
 *   0  PASSES: one sweep down the pyramid per advance with a bounded number of lock-step rounds per level (below); problems
 *      that need more are carried over to the next pass;
 *   1  TICKS: every resident problem advances ONE LM round per tick whatever level it stands on -- one evaluation launch over
 *      a device-built list of (problem, chunk) items of all levels mixed, one LM launch that steps the problems, stages the
 *      next tick's items, retires finished problems and refills their slots from the waiting list ON THE DEVICE; an advance
 *      runs ticks_per_advance ticks with one host read-back at its end (0: keep what is in force; -1: back to the stream's own
 *      choice).  Until a number is named the stream sizes every advance itself: the ticks that retire 7/8 of what the advance hands over, from the mean life (in
 *      ticks) of the problems retired so far -- the rest stays in the device's waiting ring as the buffer a freed slot is
 *      refilled from (32 until the first problems have retired; with dsm_params.fixed_schedule: one cohort's whole life).
 *      The default.
 * Both are scheduling only: results are bit-identical to the batch calls. */
int dsm_stream_set_engine(dsm_stream *s, int engine, int ticks_per_advance);
/* Tick engine, CHAINS (round 6): a problem whose pending evaluation is ONE chunk (a pyramid level of at most 4096 template points under
 * the default chunk table: every level of a semi-dense template but the finest one or two; the two coarsest levels of the metric's
 * dense pyramid) needs no other workgroup for its LM round, so the workgroup that evaluates the chunk also steps the problem -- state
 * and descriptor in LDS, the partial never leaves it -- and runs the NEXT round too, up to max_rounds rounds inside one tick, for as long
 * as the staged evaluation stays one chunk.  That is the reference's LM loop (TrackerAndScaler.cpp:505-593) running in place on the
 * levels where it is a chain of tiny steps: such a frame lives a third of the ticks.
 *   max_rounds > 0  every problem with a one-chunk evaluation chains, up to that many rounds per tick;
 *   max_rounds 0    off: every round costs a tick;
 *   max_rounds -1   the default: DSM_STREAM_CHAIN_DEFAULT rounds, for problems whose level-0 template has at most
 *                   DSM_STREAM_CHAIN_DEFAULT_MAX_N0 points.  Measured: semi-dense templates (10 k points) + 31-33 %
 *                   frames/s; dense templates gain nothing (their ticks wait for level 0's items, and a chain that runs longer than
 *                   those makes every other resident problem wait), hence the rule.
 * Scheduling only: same chunk, same partial, same reduction order -- results are bit-identical (tests/test_stream.py).  May be changed
 * between advances. */
#define DSM_STREAM_CHAIN_DEFAULT 8
#define DSM_STREAM_CHAIN_DEFAULT_MAX_N0 65536 /* the default applies to problems whose level-0 template has at most this many points */
int dsm_stream_set_chain(dsm_stream *s, int max_rounds);
/* rounds per level of a pass = this quantile (default 0.75) of the rounds recently retired problems needed there;
 * lvl < 0: every level */
int dsm_stream_set_quantile(dsm_stream *s, int lvl, double q);
/* ... or fixed: rounds_per_level[DSM_MAX_LEVELS] for problems of `mode` (0 track, 1 scale); entries <= 0 / NULL = learnt */
int dsm_stream_set_rounds(dsm_stream *s, int mode, const int *rounds_per_level);
/* CUMULATIVE statistics of the stream as of the last read-back (evaluations staged, algorithmic bytes of the problems retired,
 * launches = rounds / ticks per level, device time of the advances, per-dispatch timing as dsm_context_set_timing): callers difference them */
int dsm_stream_get_stats(dsm_stream *s, dsm_stats *track_out, dsm_stats *scale_out);
/* the schedule in force and the stream's counters: passes run, problems retired (of `mode`), slot-passes spent carrying */
int dsm_stream_get_schedule(dsm_stream *s, int mode, int *rounds_out, long long *passes_out, long long *retired_out, long long *carried_out);

/* Batched forms: n independent trackers of one context advance in lock-step launches
 * (SURVEY.md section 7 "throughput mode").  Arrays are n x 7 / n x 2 / n x DSM_MAX_LEVELS / n x 3. */
int dsm_track_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, double *pose_io, double *aff_io,
                    int coarsest_lvl, const double *min_res_for_abort, double *last_residuals,
                    double *flow_out, int *good);
int dsm_optimize_scale_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, float *scale_io,
                             int coarsest_lvl, float *err_out);

/* outputs the reference exposes as public members (TrackerAndScaler.h:59-64) */
int dsm_tracker_ref_frame_id(dsm_tracker *t);

/* Geometry of the device reduction for level lvl with n template points (tests / DESIGN.md):
 * threads per workgroup, points per thread, number of chunks. */
int dsm_reduction_geometry(dsm_tracker *t, int lvl, int n, int *threads, int *pts_per_thread,
                           int *chunks);

/* ---- loop-closure pose estimation ("next" row N2) ------------------------------------- */
/* replaces PoseEstimator::PoseEstimator(w,h) / ~PoseEstimator (PoseEstimator.cpp:41-60) */
int dsm_pose_estimator_create(dsm_context *ctx, int w, int h, int nlevels, const dsm_params *params,
                              dsm_pose_estimator **out);
int dsm_pose_estimator_destroy(dsm_pose_estimator *pe);
/* replaces PoseEstimator::estimate (PoseEstimator.h:42, PoseEstimator.cpp:298-506): direct alignment of a
 * matched keyframe's 3-D points (xyz: n x 3 doubles in the matched frame, ref_colors[lvl][i] = the
 * per-level reference intensities of LoopFrame::pts_dso, LoopHandler.cpp:172-180) into the current
 * keyframe's pyramid new_dIp with intrinsics new_cam = {fx,fy,cx,cy}.  ref_to_new_io: row-major 4x4,
 * initial guess in, result out (always written, as at :466).  *ok = aff_good && pose_error < RES_THRES(10)
 * && inlier_percent > INNER_PERCENT(90)  (:469-505). */
int dsm_pose_estimator_estimate(dsm_pose_estimator *pe, int n_pts, const double *xyz,
                                const float *const *ref_colors, float ref_ab_exposure,
                                const float *const *new_dIp, float new_ab_exposure, const float new_cam[4],
                                int coarsest_lvl, double ref_to_new_io[16], float *pose_error, int *ok);

/* ---- ring-key database (ScanContext place recognition) ------------------------------- */
/* replaces the flann::Index built at LoopHandler.cpp:35-39 plus the function-static delay
 * queue of search_ringkey (search_place.h:43-45).  dim=20, margin=LOOP_MARGIN=100,
 * k=FLANN_NN=3, thres=RINGKEY_THRES=0.1 in the reference.  dummy_key (dim floats, may be
 * NULL = zeros) is the content of the reference's uninitialised index slot 0 (quirk Q8).
 * capacity = number of keys the device shard can hold (grows by doubling when exceeded).
 * shard_rank / shard_count: this handle stores only ordinals with ordinal % count == rank. */
int dsm_ringdb_create(dsm_context *ctx, int dim, int margin, int k, float thres,
                      const float *dummy_key, int64_t capacity, int shard_rank, int shard_count,
                      dsm_ringdb **out);
int dsm_ringdb_destroy(dsm_ringdb *db);
/* number of entries in the (global) index, dummy included (flann Index::size()) */
int64_t dsm_ringdb_size(dsm_ringdb *db);
/* replaces search_ringkey (search_place.h:25-57): query, threshold, then delay-queue insert.
 * cand_out[k] receives 0..k candidate ordinals (index-1) in ascending-distance order.
 * On a sharded handle this is a COLLECTIVE call (the call site LoopHandler.cpp:247 runs once per rank with the same
 * key): needs dsm_ringdb_attach_comm; every rank scans its shard, the candidates are merged by RCCL all-reduce(min)
 * and every rank returns the same list.  Without a communicator a sharded handle fails with DSM_ERR_STATE. */
int dsm_ringdb_query_then_enqueue(dsm_ringdb *db, const float *key, int *cand_out, int *ncand_out);
/* bulk insert (bench / sharded DB): n_keys keys appended directly to the index */
int dsm_ringdb_add_points(dsm_ringdb *db, const float *keys, int64_t n_keys);
/* the delay-queue half of search_ringkey alone (search_place.h:41-56): used by sharded callers
 * that merge candidates across GPUs between the query and the insert */
int dsm_ringdb_enqueue(dsm_ringdb *db, const float *key);
/* batched exact k-NN over this shard: for each of nq queries (host pointer, nq x dim) writes k packed
 * candidates  (int64 = float_bits(dist2) << 32 | global index), ascending, into the DEVICE buffer
 * d_packed_out (nq*k int64).  Only entries with dist2 < thres are candidates; empty slots hold
 * DSM_RINGDB_NO_CANDIDATE.  The cross-shard merge is dsm_ringdb_merge_topk below. */
#define DSM_RINGDB_NO_CANDIDATE 0x7FFFFFFFFFFFFFFFll
int dsm_ringdb_knn_packed(dsm_ringdb *db, const float *queries, int nq, void *d_packed_out);
/* same, queries already on the device (nq x dim float32) */
int dsm_ringdb_knn_packed_dev(dsm_ringdb *db, const void *d_queries, int nq, void *d_packed_out);
/* same, result copied to host memory (nq*k int64) */
int dsm_ringdb_knn_packed_host(dsm_ringdb *db, const float *queries, int nq, int64_t *packed_out);

/* ---- sharded ring-key database across the GPUs of a node -------- */
/* The index of LoopHandler.cpp:35-39 split `ordinal mod G` over G processes (one per GPU); search_ringkey's k-NN
 * (search_place.h:29-33) becomes a local scan per shard + a cross-shard merge of the packed candidates over xGMI.
 * The communicator wraps an RCCL communicator (librccl.so.1 is loaded at run time, on first use):
 *   rank 0: dsm_comm_unique_id(id) -> the host distributes the 128 bytes to all ranks (its launcher's business: MPI,
 *   a file, a socket; bench.py uses torch.distributed) -> every rank: dsm_comm_create(ctx, id, rank, nranks, &comm). */
#define DSM_COMM_ID_BYTES 128
int dsm_comm_unique_id(unsigned char id_out[DSM_COMM_ID_BYTES]);
int dsm_comm_create(dsm_context *ctx, const unsigned char id[DSM_COMM_ID_BYTES], int rank, int nranks, dsm_comm **out);
int dsm_comm_destroy(dsm_comm *comm);
int dsm_comm_rank(dsm_comm *comm);
int dsm_comm_size(dsm_comm *comm);
/* Cross-shard merge, collective over the communicator: d_packed (DEVICE, nq*k int64) holds this rank's sorted local
 * candidates (dsm_ringdb_knn_packed*) on entry and the global top-k (identical on every rank, ascending) on return.
 * Enqueued on the context's stream; synchronise the context before reading the buffer from the host.
 * algo: DSM_MERGE_ALLREDUCE_MIN = k rounds of ncclAllReduce(ncclMin, ncclUint64) with winner pop (8*nq bytes per round);
 *       DSM_MERGE_ALLGATHER     = one ncclAllGather of every rank's nq*k candidates + local k-way merge.  Same result. */
enum { DSM_MERGE_ALLREDUCE_MIN = 0, DSM_MERGE_ALLGATHER = 1 };
int dsm_ringdb_merge_topk(dsm_ringdb *db, dsm_comm *comm, void *d_packed, int nq, int algo);
/* the same merge over a caller-supplied transport (hosts that do not use RCCL; tests that run the ranks as threads or
 * gloo processes): the callbacks run the collective on DEVICE buffers, in order, on `hip_stream`, and return 0 on success.
 * allreduce_min: element-wise unsigned 64-bit min over the ranks, in place.  allgather: recv[r*count .. ] = rank r's send. */
typedef int (*dsm_allreduce_min_u64_fn)(void *user, void *d_buf, size_t count, void *hip_stream);
typedef int (*dsm_allgather_u64_fn)(void *user, const void *d_send, void *d_recv, size_t count, void *hip_stream);
int dsm_ringdb_merge_topk_with(dsm_ringdb *db, void *d_packed, int nq, int algo, int nranks,
                               dsm_allreduce_min_u64_fn allreduce_min, dsm_allgather_u64_fn allgather, void *user);
/* attach (or, with NULL, detach) the communicator dsm_ringdb_query_then_enqueue uses on a sharded handle.  Borrowed:
 * dsm_comm_destroy detaches it from every database it is attached to (a later collective query on such a database fails
 * with DSM_ERR_STATE); destroy communicators before their context.  A collective query starts with an agreement round
 * (one all-reduce word per rank): if any rank cannot take part -- allocation failure, a different number of entries --
 * EVERY rank returns an error and none enters the merge rounds. */
int dsm_ringdb_attach_comm(dsm_ringdb *db, dsm_comm *comm);
/* the same with a caller-supplied transport instead of an RCCL communicator (allreduce_min NULL detaches) */
int dsm_ringdb_attach_transport(dsm_ringdb *db, int nranks, dsm_allreduce_min_u64_fn allreduce_min, dsm_allgather_u64_fn allgather,
                                void *user);

/* replaces ScanContext::generate (src/loop_closure/loop_detection/ScanContext.cpp:78-141, with
 * align_points_PCA :19-66).  Host side by design: a few 10^3 points per
 * keyframe.  pts: n x 3 doubles.  ringkey_out: num_r floats.  sig_idx_out / sig_val_out: capacity
 * num_s*num_r, *n_sig_out entries written (sparse signature, ascending bin index).  tfm_pca_rig_out:
 * row-major 4x4.  Eigenvector signs are not defined by the reference (Eigen's solver); here every
 * eigenvector is oriented so that its largest-magnitude component is positive. */
int dsm_scancontext_generate(const double *pts, int n, double lidar_range, int num_s, int num_r,
                             float *ringkey_out, int *sig_idx_out, double *sig_val_out, int *n_sig_out,
                             double *tfm_pca_rig_out);

/* replaces generate_spherical_points (loop_detection/generate_spherical_points.h:27-85; call site LoopHandler.cpp:186-187) in
 * flat-array form.  kf_pose_wc: n_kf x 6 Sophus tangents (translation, rotation) as in id_pose_wc; cur_cw: row-major 3x4
 * matrix of the current keyframe's camera<-world pose; pt_kf_id / pt_xyz: the nearby points with the keyframe that owns each.
 * Outputs: kf_keep[n_kf] (0: the reference erases that keyframe, :33-41), *n_out selected points, sel_idx (their indices in the
 * input list: the reference's updated pts_nearby) and pts_spherical (n_out x 3, current camera frame).  Per 1 x 0.5 x 1 m voxel the
 * highest point (smallest y) is kept (:64-76).  Order of the output: ascending voxel index -- the reference emits its
 * unordered_map in implementation-defined order.  Host side (tens of thousands of points per keyframe). */
int dsm_generate_spherical_points(int n_kf, const int *kf_ids, const double *kf_pose_wc, const double *cur_cw,
                                  double lidar_range, int n_pts, const int *pt_kf_id, const double *pt_xyz, int *kf_keep,
                                  int *n_out, int *sel_idx, double *pts_spherical);

/* DEVICE form of the pair generate_spherical_points + ScanContext::generate (the two calls a marginalised keyframe makes on its
 * way to the ring-key search: LoopHandler.cpp:186-187 and ScanContext.cpp:78-141 via LoopHandler.cpp:240-245), batched over
 * n_jobs keyframes -- one per concurrent sequence sharing the GPU (BASELINE configs[4]).  Per job the inputs are those of
 * dsm_generate_spherical_points; outputs: kf_keep, *n_out, sel_idx, pts_spherical as there, and -- when ringkey is not NULL --
 * ringkey[num_r], the sparse signature (sig_idx / sig_val, capacity num_s*num_r, *n_sig entries) and tfm_pca_rig[16] as
 * dsm_scancontext_generate returns them for pts_spherical.  The voxel "highest point" filter runs as two atomic-min passes
 * over a dense voxel grid plus an ordered compaction, the polar binning as atomic max per bin (csrc/loopdet_kernels.hip); the
 * keyframe trim (a handful of keyframes) and the 3x3 eigen-decomposition stay on the host.  Same results as the two host
 * entry points (bit for bit wherever device and host libm agree on atan2).  lidar_range up to 100 m (dense grid). */
typedef struct dsm_loop_job {
  int n_kf;
  const int *kf_ids;
  const double *kf_pose_wc;   /* n_kf x 6 */
  const double *cur_cw;       /* row-major 3x4 */
  int n_pts;
  const int *pt_kf_id;
  const double *pt_xyz;       /* n_pts x 3.  If pt_kf_id AND pt_xyz of a job lie in page-locked memory (dsm_host_alloc, hipHostMalloc,
                                 hipHostRegister) the device reads them in place -- no host copy of the cloud; pageable clouds are staged
                                 (28 bytes per point copied on the host).  Same results either way. */
  int *kf_keep;               /* out: n_kf */
  int *n_out;                 /* out */
  int *sel_idx;               /* out: capacity n_pts; NULL (together with pts_spherical): the selected points stay on the device */
  double *pts_spherical;      /* out: capacity n_pts x 3 */
  float *ringkey;             /* out: num_r floats, or NULL to stop after the point filter */
  int *sig_idx;               /* out: capacity num_s*num_r */
  double *sig_val;            /* out: capacity num_s*num_r */
  int *n_sig;                 /* out */
  double *tfm_pca_rig;        /* out: row-major 4x4 */
} dsm_loop_job;
int dsm_loop_descriptors_batch(dsm_context *ctx, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r);
/* The per-keyframe loop chain -- generate_spherical_points (LoopHandler.cpp:186), ScanContext::generate (:236), search_ringkey (:247,
 * search_place.h:25-57) -- as ONE enqueue and ONE read-back for the keyframes marginalised in the same advance (one per concurrent
 * sequence; n_jobs <= the index's margin): the ring keys go from the descriptor kernels to the index's k-NN on the device.  Every job
 * needs its descriptor outputs (ringkey ... tfm_pca_rig); sel_idx / pts_spherical may both be NULL (the selected points -- 0.45 MB per
 * keyframe -- then stay on the device).  Results equal the jobs' dsm_loop_descriptors_batch + dsm_ringdb_query_then_enqueue calls one
 * after the other, bit for bit: cand_out[j * k ...] / ncand_out[j] = search_ringkey's candidate list of job j, and every key is enqueued.
 * Unsharded index; num_r = its key dimension.
 * Errors are all-or-nothing: a job whose filtered cloud is empty while a descriptor is asked for fails the call with DSM_ERR_INVALID
 * BEFORE any output array of any job is written and before any key is enqueued (dsm_loop_descriptors_batch likewise writes nothing). */
int dsm_loop_detect_batch(dsm_context *ctx, dsm_ringdb *db, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r,
                          int *cand_out, int *ncand_out);

/* replaces TrackerAndScaler::makeCoarseDepthL0 (TrackerAndScaler.cpp:143-315) for callers that hold
 * the active points as flat arrays: (pu,pv) = centerProjectedTo[0..1], pidepth = centerProjectedTo[2],
 * pweight = sqrtf(1e-3/(HdiF+1e-12)) (:155-158).  ref_dIp[lvl]: the keyframe's (I,dx,dy) pyramid.
 * Outputs: n_out[lvl] and the four template lists (capacity w_l*h_l each) -- exactly the arguments of
 * dsm_tracker_set_ref.  Host side in this round (row A4 / N3). */
int dsm_make_coarse_depth_l0(int w, int h, int nlevels, int npts, const float *pu, const float *pv,
                             const float *pidepth, const float *pweight, const float *const *ref_dIp,
                             int *n_out, float *const *pc_u, float *const *pc_v, float *const *pc_idepth,
                             float *const *pc_color);

/* replaces the inner loop of search_sc (search_place.h:67-79): sparse merge-join distance of
 * two ScanContext signatures.  Host side by design (<= 3 candidates per query). */
float dsm_sc_distance(const int *sigA_idx, const double *sigA_val, int nA, const int *sigB_idx,
                      const double *sigB_val, int nB, int sc_width);
/* replaces search_sc (search_place.h:59-84) over caller supplied candidate signatures */
int dsm_search_sc(const int *sig_idx, const double *sig_val, int n_sig, int n_cand,
                  const int *cand_ids, const int *const *cand_idx, const double *const *cand_val,
                  const int *cand_n, int sc_width, int *res_idx, float *res_diff);

/* replaces the file output of LoopHandler::savePose (LoopHandler.cpp:59-80): writes n lines "incoming_id x y z" with six
 * significant digits (std::setprecision(6)) -- the dslam.txt (optimised poses, tfm_w_c.translation()) and sodso.txt
 * (trans_w_c_orig) trajectory surface of the reference.  t_wc: n x 3 doubles. */
int dsm_write_trajectory(const char *path, int n, const int *incoming_ids, const double *t_wc);

#ifdef __cplusplus
}
#endif
#endif /* DSM_HOTPATH_H */
