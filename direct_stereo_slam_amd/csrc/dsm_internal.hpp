// dsm_internal.hpp -- host-side objects behind the opaque C handles of include/dsm_hotpath.h
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/dsm_hotpath.h"
#include "dsm_kernels.hpp"

namespace dsm {
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);
} // namespace dsm

#define DSM_HIP(expr)                                                     \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) return dsm::hip_fail(e__, #expr, __FILE__, __LINE__); \
  } while (0)

struct dsm_context {
  int device = 0;
  hipStream_t stream = nullptr;
  // batched calls may split the batch over several streams so that one group's small kernels overlap
  // another group's (dsm_context_set_streams); single calls always use `stream`
  int n_streams = 1;
  std::vector<hipStream_t> extra_streams;
  std::vector<hipEvent_t> join_events;
  hipEvent_t fork_event = nullptr;
  int streams_sharing_a_queue = 0;        // streams of ensure_streams that could not be given a hardware queue of their own
  hipStream_t companion_stream = nullptr; // the companion segment of dsm_track_and_scale_batch
  hipEvent_t companion_event = nullptr;
  hipEvent_t copy_event = nullptr; // end of a host->device hand-over (dsm_tracker_upload_image)
  // descriptor updates of a batch in one copy (sync_descs): [n descriptors][n destination pointers], pinned + device
  unsigned char *h_desc_stage = nullptr, *d_desc_stage = nullptr;
  int desc_stage_cap = 0;
  hipEvent_t desc_event = nullptr;
  bool desc_stage_busy = false;
  // batched hand-over (dsm_upload_images): copies on a stream of their own, one event per group of images
  hipStream_t copy_stream = nullptr;
  // asynchronous hand-over (dsm_upload_images_async): its own work stream; copies_event = the caller's buffers are
  // free, done_event = the pyramids are built
  hipStream_t upload_stream = nullptr;
  hipEvent_t upload_copies_event = nullptr, upload_done_event = nullptr;
  bool upload_pending = false;
  bool enqueue_pending = false; // dsm_upload_images_enqueue: copy_event not yet waited for
  int async_copy_blocks = 48;  // workgroups of the host-read kernel of the asynchronous hand-over (measured: DESIGN.md section 2)
  std::vector<hipEvent_t> upload_events;
  dsm::PyrJob *d_pyr_jobs = nullptr, *h_pyr_jobs = nullptr; // h: pinned
  int pyr_jobs_cap = 0;
  // batch workspaces (grown on demand)
  int cap_prob = 0;
  int partial_stride = 0; // floats per problem
  dsm::TrackerDev **d_tracker_ptrs = nullptr;
  dsm::TrackerDev **h_tracker_ptrs = nullptr; // pinned
  dsm::LMState *d_states = nullptr;
  dsm::LMState *h_states = nullptr; // pinned
  float *d_partials = nullptr;
  dsm::StartInfo *d_start = nullptr;
  dsm::StartInfo *h_start = nullptr; // pinned
  dsm::SingleOut *d_single = nullptr;
  dsm::SingleOut *h_single = nullptr; // pinned
  int *d_status = nullptr;
  int *h_status = nullptr; // pinned
  dsm::WorkQueue *d_queue = nullptr, *h_queue = nullptr; // work-queue kernel: header (device / pinned copy)
  unsigned long long *d_qitems = nullptr;
  size_t qcap = 0;
  int queue_blocks[3] = {0, 0, 0}; // co-resident grid size per mode
  int *d_rowmap = nullptr, *h_rowmap = nullptr; // compact launches: row -> problem (relative to the segment), device / pinned
  int *d_tickets = nullptr; // per-problem arrival counters of the fused eval+LM kernels (zero between launches)
  // staging for host->device template / frame uploads
  float *d_stage = nullptr;
  size_t stage_floats = 0;
  // dsm_loop_descriptors_batch / dsm_loop_detect_batch: device arena and its page-locked mirror (grown on demand)
  void *loop_dev = nullptr, *loop_pin = nullptr;
  size_t loop_dev_bytes = 0, loop_pin_bytes = 0;
  // dsm_set_refs_from_points: page-locked mirror of the jobs' points and job table, and of their counts
  float *h_tpl_stage = nullptr;
  size_t tpl_stage_floats = 0;
  int *h_tpl_counts = nullptr;
  size_t tpl_counts_cap = 0;
  // speculative launch schedule per mode (0 = track, 1 = scale, 2 = loop-closure pose) and level, adapted after every call
  int sched[3][DSM_MAX_LEVELS] = {{6, 8, 10, 12, 16, 16}, {4, 4, 4, 4, 4, 4}, {6, 8, 10, 12, 16, 16}};
  // ... and the number of launches after which only a level's stragglers are still at work (the third quartile of the
  // rounds the problems of recent calls needed): from there on a round is ONE fused evaluate + step launch (dsm_params.fuse_lm = 1)
  int sched_bulk[3][DSM_MAX_LEVELS] = {{1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30}, {1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30},
                                       {1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30, 1 << 30}};
  int sched_bulk_key = -1; // batch-size bucket and launch form the sched_bulk figures were learnt on
  // stats / timing
  bool timing = false;
  dsm_stats stats{};
  dsm_stats stats2{}; // companion segment of the last dsm_track_and_scale_batch
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t ev_total[2] = {nullptr, nullptr};
};

struct dsm_tracker {
  dsm_context *ctx = nullptr;
  int w = 0, h = 0, nlevels = 0;
  dsm_params params{};
  dsm::TrackerDev desc{}; // host copy of the device descriptor
  dsm::TrackerDev *d_desc = nullptr;
  float4 *d_pts[DSM_MAX_LEVELS] = {};
  int pts_cap[DSM_MAX_LEVELS] = {}; // template capacity per level (w_l*h_l; w*h on every level for the pose estimator)
  float *d_img[2][DSM_MAX_LEVELS] = {};
  float *d_raw[2] = {nullptr, nullptr}; // raw level-0 images of dsm_tracker_upload_image, per slot
  // back buffers of the two frame slots (DSM_SLOT_NEXT_*), swapped in by dsm_frames_advance
  float *d_img_back[2][DSM_MAX_LEVELS] = {};
  float *d_raw_back[2] = {nullptr, nullptr};
  bool have_back[2] = {false, false};
  float back_exposure[2] = {1.f, 1.f};
  bool have_k = false, have_ref = false, have_frame[2] = {false, false};
  int ref_frame_id = -1;
  bool desc_dirty = true;
};

namespace dsm {
int ensure_batch_capacity(dsm_context *ctx, int nprob, int partial_stride);
int ensure_stage(dsm_context *ctx, size_t floats);
int sync_desc(dsm_tracker *t);
int sync_descs(dsm_context *ctx, dsm_tracker *const *ts, int n);
int check_ready(dsm_tracker *t, int mode);
hipEvent_t get_event(dsm_context *ctx, size_t idx);
int ensure_streams(dsm_context *ctx, int ng, bool companion);
void collect_eval_timing(dsm_context *ctx, const std::vector<int> &ev_lvl, int nlevels, dsm_stats &st);
} // namespace dsm
