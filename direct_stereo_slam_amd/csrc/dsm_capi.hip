// dsm_capi.hip -- C ABI (include/dsm_hotpath.h) of the tracker / scale-optimiser hot path.
// Host side only orchestrates: uploads, launch sequencing of the LM state machine, read-back.
// All arithmetic of the path runs in the HIP kernels of tracker_kernels.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "dsm_internal.hpp"
#include <algorithm>
#include <utility>

namespace dsm {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  g_err = buf;
  return DSM_ERR_HIP;
}

static int invalid(const char *msg) {
  set_error(msg);
  return DSM_ERR_INVALID;
}

// One intensity plane of a w x h level.  Lanes whose point is not usable fetch their twelve taps around texel (2, 2)
// instead (rows 1-4, columns 1-4), always inside the image (dsm_tracker_create wants a coarsest level of at least 8 x 8);
// the four rows and few texels of slack are belt and braces.
static size_t plane_bytes(int w, int h) { return sizeof(float) * ((size_t)w * h + 4 * (size_t)w + 16); }

int ensure_stage(dsm_context *ctx, size_t floats) {
  if (floats <= ctx->stage_floats) return DSM_OK;
  if (ctx->d_stage) DSM_HIP(hipFree(ctx->d_stage));
  ctx->d_stage = nullptr;
  ctx->stage_floats = 0;
  DSM_HIP(hipMalloc(&ctx->d_stage, floats * sizeof(float)));
  ctx->stage_floats = floats;
  return DSM_OK;
}

template <typename T>
static int realloc_dev(T **p, size_t n) {
  if (*p) DSM_HIP(hipFree(*p));
  *p = nullptr;
  DSM_HIP(hipMalloc(p, n * sizeof(T)));
  return DSM_OK;
}
template <typename T>
static int realloc_pinned(T **p, size_t n) {
  if (*p) DSM_HIP(hipHostFree(*p));
  *p = nullptr;
  DSM_HIP(hipHostMalloc(p, n * sizeof(T), hipHostMallocDefault));
  return DSM_OK;
}

int ensure_batch_capacity(dsm_context *ctx, int nprob, int partial_stride) {
  if (nprob <= ctx->cap_prob && partial_stride <= ctx->partial_stride) return DSM_OK;
  const int cap = nprob > ctx->cap_prob ? nprob : ctx->cap_prob;
  const int ps = partial_stride > ctx->partial_stride ? partial_stride : ctx->partial_stride;
  int rc;
  if ((rc = realloc_dev(&ctx->d_tracker_ptrs, cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_tracker_ptrs, cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_states, cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_states, cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_partials, (size_t)cap * ps))) return rc;
  if ((rc = realloc_dev(&ctx->d_start, cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_start, cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_single, cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_single, cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_status, 2 * (size_t)cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_status, 2 * (size_t)cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_rowmap, cap))) return rc;
  if ((rc = realloc_pinned(&ctx->h_rowmap, cap))) return rc;
  if ((rc = realloc_dev(&ctx->d_tickets, cap))) return rc;
  DSM_HIP(hipMemsetAsync(ctx->d_tickets, 0, sizeof(int) * cap, ctx->stream));
  DSM_HIP(hipMemsetAsync(ctx->d_states, 0, sizeof(LMState) * cap, ctx->stream));
  ctx->cap_prob = cap;
  ctx->partial_stride = ps;
  return DSM_OK;
}

int sync_desc(dsm_tracker *t) {
  if (!t->desc_dirty) return DSM_OK;
  DSM_HIP(hipMemcpyAsync(t->d_desc, &t->desc, sizeof(TrackerDev), hipMemcpyHostToDevice, t->ctx->stream));
  // the host copy may change again before the copy engine reads it
  DSM_HIP(hipStreamSynchronize(t->ctx->stream));
  t->desc_dirty = false;
  return DSM_OK;
}

// the same for a batch: all changed descriptors through one pinned staging buffer, one copy and one scatter launch
int sync_descs(dsm_context *ctx, dsm_tracker *const *ts, int n) {
  int dirty = 0;
  for (int i = 0; i < n; i++) dirty += ts[i]->desc_dirty ? 1 : 0;
  if (dirty <= 2) {
    for (int i = 0; i < n; i++) {
      int rc = sync_desc(ts[i]);
      if (rc) return rc;
    }
    return DSM_OK;
  }
  const size_t per = sizeof(TrackerDev) + sizeof(TrackerDev *);
  if (dirty > ctx->desc_stage_cap) {
    if (ctx->desc_stage_busy) DSM_HIP(hipEventSynchronize(ctx->desc_event));
    ctx->desc_stage_busy = false;
    int rc = realloc_dev(&ctx->d_desc_stage, per * dirty);
    if (rc) return rc;
    rc = realloc_pinned(&ctx->h_desc_stage, per * dirty);
    if (rc) return rc;
    ctx->desc_stage_cap = dirty;
  }
  if (!ctx->desc_event) DSM_HIP(hipEventCreateWithFlags(&ctx->desc_event, hipEventDisableTiming));
  if (ctx->desc_stage_busy) DSM_HIP(hipEventSynchronize(ctx->desc_event)); // the previous copy still reads the buffer
  TrackerDev *hd = (TrackerDev *)ctx->h_desc_stage;
  TrackerDev **hp = (TrackerDev **)(ctx->h_desc_stage + sizeof(TrackerDev) * dirty);
  int k = 0;
  for (int i = 0; i < n; i++) {
    if (!ts[i]->desc_dirty) continue;
    bool seen = false; // the same tracker twice in a batch (hypotheses)
    for (int j = 0; j < k && !seen; j++) seen = hp[j] == ts[i]->d_desc;
    if (seen) continue;
    hd[k] = ts[i]->desc;
    hp[k] = ts[i]->d_desc;
    k++;
  }
  DSM_HIP(hipMemcpyAsync(ctx->d_desc_stage, ctx->h_desc_stage, per * dirty, hipMemcpyHostToDevice, ctx->stream));
  launch_desc_scatter(ctx->stream, k, (const TrackerDev *)ctx->d_desc_stage,
                      (TrackerDev *const *)(ctx->d_desc_stage + sizeof(TrackerDev) * dirty));
  DSM_HIP(hipGetLastError());
  DSM_HIP(hipEventRecord(ctx->desc_event, ctx->stream));
  ctx->desc_stage_busy = true;
  for (int i = 0; i < n; i++) ts[i]->desc_dirty = false;
  return DSM_OK;
}

// ---- small host math (float, contraction off): makeK, TrackerAndScaler.cpp:117-141 ----
static float cof3(const float *m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void mat3f_inverse(const float *m, float *r) { // Eigen Matrix3f::inverse(), cofactor form
  const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  const float invdet = 1.0f / det;
  r[0] = c0 * invdet;
  r[1] = c1 * invdet;
  r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet;
  r[4] = cof3(m, 1, 1) * invdet;
  r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet;
  r[7] = cof3(m, 1, 2) * invdet;
  r[8] = cof3(m, 2, 2) * invdet;
}

// SE3(Matrix4d) of Sophus/Eigen for tfm_f1_f0_ (TrackerAndScaler.cpp:82-86)
static void se3_from_matrix(const double T[16], double pose[7]) {
  const double M[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
  double q[4];
  double t = M[0][0] + M[1][1] + M[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M[2][1] - M[1][2]) * t;
    q[1] = (M[0][2] - M[2][0]) * t;
    q[2] = (M[1][0] - M[0][1]) * t;
  } else {
    int i = 0;
    if (M[1][1] > M[0][0]) i = 1;
    if (M[2][2] > M[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(M[i][i] - M[j][j] - M[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M[k][j] - M[j][k]) * t;
    q[j] = (M[j][i] + M[i][j]) * t;
    q[k] = (M[k][i] + M[i][k]) * t;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) pose[i] = q[i] / n;
  pose[4] = T[3];
  pose[5] = T[7];
  pose[6] = T[11];
}

static int round8(int x) { return (x + 7) & ~7; }

// record an event pair around a launch when timing is on
struct EvSlot {
  hipEvent_t a, b;
  int lvl;
};

} // namespace dsm

using namespace dsm;

extern "C" {

const char *dsm_last_error(void) { return g_err.c_str(); }
int dsm_abi_version(void) { return DSM_ABI_VERSION; }

int dsm_params_default_sized(dsm_params *p, size_t caller_size) {
  if (!p || caller_size < sizeof(size_t)) return invalid("dsm_params_default_sized: null struct or a size below its first member");
  dsm_params d;
  dsm_params_default(&d);
  memcpy(p, &d, caller_size < sizeof d ? caller_size : sizeof d); // never past the end of the caller's struct
  p->struct_size = caller_size;
  if (caller_size != sizeof d) {
    set_error("dsm_params: the caller was built against another version of dsm_hotpath.h (sizeof(dsm_params) differs)");
    return DSM_ERR_INVALID;
  }
  return DSM_OK;
}

void dsm_params_default(dsm_params *p) {
  memset(p, 0, sizeof *p);
  p->struct_size = sizeof(dsm_params);
  p->huber_th = 9.0f;
  p->coarse_cutoff_th = 20.0f;
  p->scale_xi_rot = 1.0f;
  p->scale_xi_trans = 0.5f;
  p->scale_a = 10.0f;
  p->scale_b = 1000.0f;
  p->affine_opt_mode_a = 0.0f;
  p->affine_opt_mode_b = 0.0f;
  p->lambda_extrapolation_limit = 0.001f;
  const int it[DSM_MAX_LEVELS] = {10, 20, 50, 50, 50, 50};
  memcpy(p->max_iterations, it, sizeof it);
  p->adaptive_schedule = 1;
  p->persistent_coarse = 0;
  p->fuse_lm = 1;
  p->work_queue = 1;
  p->speculate = 1;
  p->compact_tail = 1;
  p->fixed_schedule = 0;
  p->chunk_geometry = 0;
  p->frame_check = 1;
  p->frame_grad_tol = 0.0f;
}

int dsm_context_create(int device_ordinal, dsm_context **out) {
  if (!out) return invalid("dsm_context_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: this library has no CPU fallback");
    return DSM_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= ndev) return invalid("dsm_context_create: bad device ordinal");
  DSM_HIP(hipSetDevice(device_ordinal));
  dsm_context *ctx = new dsm_context();
  ctx->device = device_ordinal;
  hipError_t ce = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (ce == hipSuccess) ce = hipEventCreate(&ctx->ev_total[0]);
  if (ce == hipSuccess) ce = hipEventCreate(&ctx->ev_total[1]);
  if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming);
  if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ctx->copy_event, hipEventDisableTiming);
  if (ce != hipSuccess) { // give back what was created
    dsm_context_destroy(ctx);
    DSM_HIP(ce);
  }
  *out = ctx;
  return DSM_OK;
}

int dsm_context_destroy(dsm_context *ctx) {
  if (!ctx) return DSM_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  hipFree(ctx->d_tracker_ptrs);
  hipHostFree(ctx->h_tracker_ptrs);
  hipFree(ctx->d_states);
  hipHostFree(ctx->h_states);
  hipFree(ctx->d_partials);
  hipFree(ctx->d_start);
  hipHostFree(ctx->h_start);
  hipFree(ctx->d_single);
  hipHostFree(ctx->h_single);
  hipFree(ctx->d_status);
  hipFree(ctx->d_rowmap);
  if (ctx->h_rowmap) hipHostFree(ctx->h_rowmap);
  hipFree(ctx->d_tickets);
  hipFree(ctx->d_queue);
  hipFree(ctx->d_qitems);
  hipHostFree(ctx->h_queue);
  hipHostFree(ctx->h_status);
  hipFree(ctx->d_stage);
  if (ctx->h_tpl_stage) hipHostFree(ctx->h_tpl_stage);
  hipFree(ctx->loop_dev);
  if (ctx->loop_pin) hipHostFree(ctx->loop_pin);
  if (ctx->h_tpl_counts) hipHostFree(ctx->h_tpl_counts);
  for (hipEvent_t ev : ctx->ev_pool) hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->join_events) hipEventDestroy(ev);
  if (ctx->companion_stream) hipStreamDestroy(ctx->companion_stream);
  if (ctx->companion_event) hipEventDestroy(ctx->companion_event);
  for (hipStream_t st : ctx->extra_streams) hipStreamDestroy(st);
  if (ctx->fork_event) hipEventDestroy(ctx->fork_event);
  if (ctx->copy_event) hipEventDestroy(ctx->copy_event);
  for (hipEvent_t ev : ctx->upload_events) hipEventDestroy(ev);
  if (ctx->desc_event) hipEventDestroy(ctx->desc_event);
  hipFree(ctx->d_desc_stage);
  hipHostFree(ctx->h_desc_stage);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->upload_stream) {
    hipStreamSynchronize(ctx->upload_stream);
    hipStreamDestroy(ctx->upload_stream);
    hipEventDestroy(ctx->upload_copies_event);
    hipEventDestroy(ctx->upload_done_event);
  }
  hipFree(ctx->d_pyr_jobs);
  hipHostFree(ctx->h_pyr_jobs);
  for (hipEvent_t ev : ctx->ev_total)
    if (ev) hipEventDestroy(ev);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return DSM_OK;
}

int dsm_context_sync(dsm_context *ctx) {
  if (!ctx) return invalid("null context");
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}
int dsm_context_set_timing(dsm_context *ctx, int enable) {
  if (!ctx) return invalid("null context");
  ctx->timing = enable != 0;
  return DSM_OK;
}
int dsm_context_set_streams(dsm_context *ctx, int n_streams) {
  if (!ctx || n_streams < 1 || n_streams > 16) return invalid("dsm_context_set_streams: 1..16");
  ctx->n_streams = n_streams;
  return DSM_OK;
}
int dsm_context_get_stats(dsm_context *ctx, dsm_stats *out) {
  if (!ctx || !out) return invalid("null argument");
  *out = ctx->stats;
  return DSM_OK;
}
void *dsm_context_stream(dsm_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int dsm_context_stream_queues(dsm_context *ctx, int *streams_out, int *sharing_out) {
  if (!ctx) return invalid("dsm_context_stream_queues: null context");
  if (streams_out) *streams_out = 1 + (int)ctx->extra_streams.size() + (ctx->companion_stream ? 1 : 0);
  if (sharing_out) *sharing_out = ctx->streams_sharing_a_queue;
  return DSM_OK;
}

// ---- tracker ------------------------------------------------------------------------------
static int tracker_create_fill(dsm_tracker *t, dsm_context *ctx, int w, int h, int nlevels, const double T_f1_f0[16],
                               const float K1[4], const dsm_params *params);

int dsm_tracker_create(dsm_context *ctx, int w, int h, int nlevels, const double T_f1_f0[16],
                       const float K1[4], const dsm_params *params, dsm_tracker **out) {
  if (!ctx || !out || !T_f1_f0 || !K1) return invalid("dsm_tracker_create: null argument");
  if (nlevels < 1 || nlevels > DSM_MAX_LEVELS) return invalid("dsm_tracker_create: nlevels out of range");
  if ((w >> (nlevels - 1)) < 8 || (h >> (nlevels - 1)) < 8) return invalid("dsm_tracker_create: image too small for nlevels");
  // the eval kernels address texels with 32-bit byte offsets (16 bytes per texel at most)
  if ((long long)w * h * 16 >= (1ll << 32)) return invalid("dsm_tracker_create: image too large");
  DSM_HIP(hipSetDevice(ctx->device));
  dsm_tracker *t = new dsm_tracker();
  const int rc = tracker_create_fill(t, ctx, w, h, nlevels, T_f1_f0, K1, params);
  if (rc != DSM_OK) { // e.g. out of device memory half way: give back what was allocated (hipFree(nullptr) is a no-op)
    const std::string keep = g_err;
    dsm_tracker_destroy(t);
    g_err = keep;
    return rc;
  }
  *out = t;
  return DSM_OK;
}

static int tracker_create_fill(dsm_tracker *t, dsm_context *ctx, int w, int h, int nlevels, const double T_f1_f0[16],
                               const float K1[4], const dsm_params *params) {
  t->ctx = ctx;
  t->w = w;
  t->h = h;
  t->nlevels = nlevels;
  if (params) {
    if (params->struct_size != sizeof(dsm_params))
      return invalid("dsm_params.struct_size does not match this library's dsm_params: the caller was built against another "
                     "version of dsm_hotpath.h (use dsm_params_default, compare dsm_abi_version() with DSM_ABI_VERSION)");
    if (params->chunk_geometry < 0 || params->chunk_geometry > 2)
      return invalid("dsm_params.chunk_geometry: 0 (throughput table), 1 (latency table) or 2 (latency table, one chunk up to 4096 points)");
    t->params = *params;
  } else {
    dsm_params_default(&t->params);
  }
  TrackerDev &D = t->desc;
  memset(&D, 0, sizeof D);
  D.nlevels = nlevels;
  D.p.huber_th = t->params.huber_th;
  D.p.coarse_cutoff_th = t->params.coarse_cutoff_th;
  D.p.scale_xi_rot = t->params.scale_xi_rot;
  D.p.scale_xi_trans = t->params.scale_xi_trans;
  D.p.scale_a = t->params.scale_a;
  D.p.scale_b = t->params.scale_b;
  D.p.affine_opt_mode_a = t->params.affine_opt_mode_a;
  D.p.affine_opt_mode_b = t->params.affine_opt_mode_b;
  D.p.lambda_extrapolation_limit = t->params.lambda_extrapolation_limit;
  for (int l = 0; l < DSM_MAX_LEVELS; l++) D.p.max_iterations[l] = t->params.max_iterations[l];
  D.p.fixed_schedule = t->params.fixed_schedule;
  D.p.geometry = t->params.chunk_geometry;
  se3_from_matrix(T_f1_f0, D.T10);
  for (int l = 0; l < nlevels; l++) { // TrackerAndScaler.cpp:52-64
    const int wl = w >> l, hl = h >> l;
    D.lv[l].w = wl;
    D.lv[l].h = hl;
    DSM_HIP(hipMalloc(&t->d_pts[l], sizeof(float4) * ((size_t)wl * hl + kTemplatePad)));
    DSM_HIP(hipMemsetAsync(t->d_pts[l], 0, sizeof(float4) * ((size_t)wl * hl + kTemplatePad), ctx->stream)); // (not the null stream: streams_overlap)
    t->pts_cap[l] = wl * hl;
    for (int s = 0; s < 2; s++) {
      DSM_HIP(hipMalloc(&t->d_img[s][l], plane_bytes(wl, hl)));
      DSM_HIP(hipMemsetAsync(t->d_img[s][l], 0, plane_bytes(wl, hl), ctx->stream));
      D.lv[l].img[s] = t->d_img[s][l];
    }
    D.lv[l].pts = t->d_pts[l];
    D.lv[l].n = 0;
  }
  // camera 1 pyramid, :89-98
  D.lv[0].fx1 = K1[0];
  D.lv[0].fy1 = K1[1];
  D.lv[0].cx1 = K1[2];
  D.lv[0].cy1 = K1[3];
  for (int l = 1; l < nlevels; l++) {
    D.lv[l].fx1 = D.lv[l - 1].fx1 * 0.5;
    D.lv[l].fy1 = D.lv[l - 1].fy1 * 0.5;
    D.lv[l].cx1 = (D.lv[0].cx1 + 0.5) / ((int)1 << l) - 0.5;
    D.lv[l].cy1 = (D.lv[0].cy1 + 0.5) / ((int)1 << l) - 0.5;
  }
  DSM_HIP(hipMalloc(&t->d_desc, sizeof(TrackerDev)));
  t->desc_dirty = true;
  DSM_HIP(hipStreamSynchronize(ctx->stream)); // the zero fills are through before any other stream may write these buffers
  return DSM_OK;
}

int dsm_tracker_destroy(dsm_tracker *t) {
  if (!t) return DSM_OK;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  if (t->ctx->upload_stream) hipStreamSynchronize(t->ctx->upload_stream); // an asynchronous hand-over may still write its buffers
  if (t->ctx->copy_stream) hipStreamSynchronize(t->ctx->copy_stream);
  for (int l = 0; l < t->nlevels; l++) {
    hipFree(t->d_pts[l]);
    hipFree(t->d_img[0][l]);
    hipFree(t->d_img[1][l]);
  }
  hipFree(t->d_raw[0]);
  hipFree(t->d_raw[1]);
  for (int s = 0; s < 2; s++) {
    hipFree(t->d_raw_back[s]);
    for (int l = 0; l < DSM_MAX_LEVELS; l++) hipFree(t->d_img_back[s][l]);
  }
  hipFree(t->d_desc);
  delete t;
  return DSM_OK;
}

int dsm_tracker_make_k(dsm_tracker *t, float fx, float fy, float cx, float cy) {
  if (!t) return invalid("null tracker");
  TrackerDev &D = t->desc;
  D.lv[0].fx = fx;
  D.lv[0].fy = fy;
  D.lv[0].cx = cx;
  D.lv[0].cy = cy;
  for (int l = 1; l < t->nlevels; l++) { // :126-133
    D.lv[l].fx = D.lv[l - 1].fx * 0.5;
    D.lv[l].fy = D.lv[l - 1].fy * 0.5;
    D.lv[l].cx = (D.lv[0].cx + 0.5) / ((int)1 << l) - 0.5;
    D.lv[l].cy = (D.lv[0].cy + 0.5) / ((int)1 << l) - 0.5;
  }
  for (int l = 0; l < t->nlevels; l++) { // :135-140
    const float K[9] = {D.lv[l].fx, 0.0f, D.lv[l].cx, 0.0f, D.lv[l].fy, D.lv[l].cy, 0.0f, 0.0f, 1.0f};
    mat3f_inverse(K, D.lv[l].Ki);
  }
  t->have_k = true;
  t->desc_dirty = true;
  return DSM_OK;
}

int dsm_tracker_set_ref(dsm_tracker *t, int ref_frame_id, double ref_aff_a, double ref_aff_b,
                        float ref_exposure, const int *n, const float *const *pc_u,
                        const float *const *pc_v, const float *const *pc_idepth,
                        const float *const *pc_color) {
  if (!t || !n || !pc_u || !pc_v || !pc_idepth || !pc_color) return invalid("dsm_tracker_set_ref: null argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  for (int l = 0; l < t->nlevels; l++)
    if (n[l] < 0 || n[l] > t->pts_cap[l]) return invalid("dsm_tracker_set_ref: n[lvl] out of range");
  int rc = ensure_stage(ctx, 4 * (size_t)t->w * t->h);
  if (rc) return rc;
  for (int l = 0; l < t->nlevels; l++) {
    const size_t nl = n[l];
    float *su = ctx->d_stage, *sv = su + nl, *si = sv + nl, *sc = si + nl;
    if (nl) {
      DSM_HIP(hipMemcpyAsync(su, pc_u[l], nl * 4, hipMemcpyHostToDevice, ctx->stream));
      DSM_HIP(hipMemcpyAsync(sv, pc_v[l], nl * 4, hipMemcpyHostToDevice, ctx->stream));
      DSM_HIP(hipMemcpyAsync(si, pc_idepth[l], nl * 4, hipMemcpyHostToDevice, ctx->stream));
      DSM_HIP(hipMemcpyAsync(sc, pc_color[l], nl * 4, hipMemcpyHostToDevice, ctx->stream));
      launch_interleave_template(ctx->stream, (int)nl, su, sv, si, sc, t->d_pts[l]);
    }
    DSM_HIP(hipStreamSynchronize(ctx->stream)); // staging buffer is reused by the next level
    t->desc.lv[l].n = (int)nl;
  }
  t->desc.ref_a = ref_aff_a; // :323-324
  t->desc.ref_b = ref_aff_b;
  t->desc.ref_exposure = ref_exposure;
  t->ref_frame_id = ref_frame_id;
  t->have_ref = true;
  t->desc_dirty = true;
  return DSM_OK;
}

int dsm_set_refs_from_points(dsm_context *ctx, int n_jobs, const dsm_ref_job *jobs) {
  if (!ctx || n_jobs < 0 || (n_jobs && !jobs)) return invalid("dsm_set_refs_from_points: bad argument");
  if (!n_jobs) return DSM_OK;
  DSM_HIP(hipSetDevice(ctx->device));
  std::vector<size_t> off((size_t)n_jobs + 1, 0);
  for (int j = 0; j < n_jobs; j++) {
    const dsm_ref_job &J = jobs[j];
    dsm_tracker *t = J.t, *fo = J.frame_owner;
    if (!t || !fo || J.slot < 0 || J.slot > 1 || J.npts < 0 || (J.npts > 0 && (!J.pu || !J.pv || !J.pidepth || !J.pweight)))
      return invalid("dsm_tracker_set_ref_from_points: bad argument");
    if (t->ctx != ctx || fo->ctx != ctx || fo->w != t->w || fo->h != t->h || fo->nlevels != t->nlevels)
      return invalid("dsm_tracker_set_ref_from_points: the frame owner must share context, size and levels");
    if (!fo->have_frame[J.slot]) {
      set_error("dsm_tracker_set_ref_from_points: the keyframe's pyramid has not been uploaded to that slot");
      return DSM_ERR_STATE;
    }
    for (int l = 0; l < t->nlevels; l++)
      if ((long long)((t->w >> l) - 4) * ((t->h >> l) - 4) > t->pts_cap[l]) return invalid("template capacity too small");
    for (int k = 0; k < j; k++)
      if (jobs[k].t == t) return invalid("dsm_set_refs_from_points: a tracker takes one reference per call");
    const size_t floats = make_coarse_depth_workspace_floats(t->w, t->h, t->nlevels, J.npts);
    off[j + 1] = off[j] + ((floats + 63) & ~(size_t)63);
  }
  // ONE launch sequence for all jobs (template_kernels.hip: blockIdx.y = job), ONE host->device copy for their points and the job table
  // (staged in page-locked memory), ONE copy back for their counts.  Device layout of the staging area: [workspaces | points | counts |
  // job table]; the pinned mirror holds [points | job table] and, separately, the counts.
  dsm_tracker *t0 = jobs[0].t;
  for (int j = 1; j < n_jobs; j++)
    if (jobs[j].t->w != t0->w || jobs[j].t->h != t0->h || jobs[j].t->nlevels != t0->nlevels)
      return invalid("dsm_set_refs_from_points: trackers of different geometry in one call");
  std::vector<size_t> poff((size_t)n_jobs + 1, 0);
  int max_npts = 0;
  for (int j = 0; j < n_jobs; j++) {
    poff[j + 1] = poff[j] + ((4 * (size_t)jobs[j].npts + 3) & ~(size_t)3);
    if (jobs[j].npts > max_npts) max_npts = jobs[j].npts;
  }
  const size_t ws_floats = off[n_jobs], pt_floats = poff[n_jobs], cnt_floats = ((size_t)n_jobs * (DSM_MAX_LEVELS + 2) + 3) & ~(size_t)3;
  const size_t tab_floats = (sizeof(TplJob) * (size_t)n_jobs + 3) / 4;
  int rc = ensure_stage(ctx, ws_floats + pt_floats + cnt_floats + tab_floats + 16);
  if (rc) return rc;
  const size_t host_floats = pt_floats + tab_floats + 16;
  if (host_floats > ctx->tpl_stage_floats) {
    if (ctx->h_tpl_stage) DSM_HIP(hipHostFree(ctx->h_tpl_stage));
    ctx->h_tpl_stage = nullptr, ctx->tpl_stage_floats = 0;
    DSM_HIP(hipHostMalloc(&ctx->h_tpl_stage, sizeof(float) * host_floats * 2, hipHostMallocDefault));
    ctx->tpl_stage_floats = host_floats * 2;
  }
  if ((size_t)n_jobs * (DSM_MAX_LEVELS + 2) > ctx->tpl_counts_cap) {
    if (ctx->h_tpl_counts) DSM_HIP(hipHostFree(ctx->h_tpl_counts));
    ctx->h_tpl_counts = nullptr, ctx->tpl_counts_cap = 0;
    DSM_HIP(hipHostMalloc(&ctx->h_tpl_counts, sizeof(int) * (size_t)n_jobs * (DSM_MAX_LEVELS + 2) * 2, hipHostMallocDefault));
    ctx->tpl_counts_cap = (size_t)n_jobs * (DSM_MAX_LEVELS + 2) * 2;
  }
  float *d_ws = ctx->d_stage, *d_pt = d_ws + ws_floats;
  int *d_cnt = (int *)(d_pt + pt_floats);
  TplJob *d_tab = (TplJob *)((float *)d_cnt + cnt_floats);
  float *h_pt = ctx->h_tpl_stage;
  TplJob *h_tab = (TplJob *)(h_pt + pt_floats);
  for (int j = 0; j < n_jobs; j++) {
    const dsm_ref_job &J = jobs[j];
    dsm_tracker *t = J.t;
    t->have_ref = false; // (until its counts are back)
    const size_t np_ = (size_t)J.npts;
    float *hp = h_pt + poff[j];
    if (np_) {
      memcpy(hp, J.pu, sizeof(float) * np_);
      memcpy(hp + np_, J.pv, sizeof(float) * np_);
      memcpy(hp + 2 * np_, J.pidepth, sizeof(float) * np_);
      memcpy(hp + 3 * np_, J.pweight, sizeof(float) * np_);
    }
    TplJob &T = h_tab[j];
    memset(&T, 0, sizeof T);
    T.npts = J.npts;
    T.pt = d_pt + poff[j];
    T.ws = d_ws + off[j];
    T.d_n = d_cnt + (size_t)j * (DSM_MAX_LEVELS + 2);
    for (int l = 0; l < t->nlevels; l++) T.ref[l] = J.frame_owner->d_img[J.slot][l], T.pts[l] = t->d_pts[l];
  }
  // (points and job table are contiguous in the pinned mirror and on the device up to the counts in between: two copies)
  if (pt_floats) DSM_HIP(hipMemcpyAsync(d_pt, h_pt, sizeof(float) * pt_floats, hipMemcpyHostToDevice, ctx->stream));
  DSM_HIP(hipMemcpyAsync(d_tab, h_tab, sizeof(TplJob) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
  launch_make_coarse_depth(ctx->stream, t0->w, t0->h, t0->nlevels, d_tab, n_jobs, max_npts, kTexel);
  DSM_HIP(hipGetLastError());
  int *h_cnt = ctx->h_tpl_counts;
  DSM_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(int) * (size_t)n_jobs * (DSM_MAX_LEVELS + 2), hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int> h_n((size_t)n_jobs * (DSM_MAX_LEVELS + 1), 0);
  for (int j = 0; j < n_jobs; j++)
    for (int l = 0; l <= jobs[j].t->nlevels; l++) h_n[(size_t)j * (DSM_MAX_LEVELS + 1) + l] = h_cnt[(size_t)j * (DSM_MAX_LEVELS + 2) + l];
  for (int j = 0; j < n_jobs; j++)
    if (h_n[(size_t)j * (DSM_MAX_LEVELS + 1) + jobs[j].t->nlevels]) // a point projected outside the image: the reference would corrupt memory (:160)
      return invalid("dsm_tracker_set_ref_from_points: point outside the level-0 image");
  for (int j = 0; j < n_jobs; j++) {
    const dsm_ref_job &J = jobs[j];
    dsm_tracker *t = J.t;
    const int *hn = h_n.data() + (size_t)j * (DSM_MAX_LEVELS + 1);
    for (int l = 0; l < t->nlevels; l++) {
      t->desc.lv[l].n = hn[l];
      if (J.n_out) J.n_out[l] = hn[l];
    }
    t->desc.ref_a = J.ref_aff_a; // :323-324
    t->desc.ref_b = J.ref_aff_b;
    t->desc.ref_exposure = J.ref_exposure;
    t->ref_frame_id = J.ref_frame_id;
    t->have_ref = true;
    t->desc_dirty = true;
  }
  return DSM_OK;
}

int dsm_tracker_set_ref_from_points(dsm_tracker *t, dsm_tracker *frame_owner, int slot, int ref_frame_id, double ref_aff_a,
                                    double ref_aff_b, float ref_exposure, int npts, const float *pu, const float *pv,
                                    const float *pidepth, const float *pweight, int *n_out) {
  if (!t) return invalid("dsm_tracker_set_ref_from_points: bad argument");
  dsm_ref_job J;
  J.t = t, J.frame_owner = frame_owner, J.slot = slot, J.ref_frame_id = ref_frame_id, J.ref_aff_a = ref_aff_a, J.ref_aff_b = ref_aff_b;
  J.ref_exposure = ref_exposure, J.npts = npts, J.pu = pu, J.pv = pv, J.pidepth = pidepth, J.pweight = pweight, J.n_out = n_out;
  return dsm_set_refs_from_points(t->ctx, 1, &J);
}

int dsm_tracker_scale_depth(dsm_tracker *t, float scale) {
  if (!t) return invalid("null tracker");
  if (!t->have_ref) {
    set_error("dsm_tracker_scale_depth before set_ref");
    return DSM_ERR_STATE;
  }
  DSM_HIP(hipSetDevice(t->ctx->device));
  // one launch over all levels, enqueued on the context's stream and NOT waited for: everything that reads the template afterwards --
  // evaluations, the streaming form's advances (their stream groups fork from this stream), dsm_tracker_get_template -- is ordered
  // behind it there (eight keyframes' calls were 0.24 of the 1.9 ms between two advances of 128 concurrent sequences)
  ScaleDepthArgs a;
  a.nlevels = t->nlevels;
  int max_n = 0;
  for (int l = 0; l < t->nlevels; l++) {
    a.n[l] = t->desc.lv[l].n, a.pts[l] = t->d_pts[l];
    if (a.n[l] > max_n) max_n = a.n[l];
  }
  launch_scale_depth_levels(t->ctx->stream, a, max_n, scale);
  DSM_HIP(hipGetLastError());
  return DSM_OK;
}

int dsm_tracker_get_template(dsm_tracker *t, int lvl, int *n, float *pc_u, float *pc_v, float *pc_idepth,
                             float *pc_color) {
  if (!t || !n || lvl < 0 || lvl >= t->nlevels) return invalid("dsm_tracker_get_template: bad argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  const size_t nl = t->desc.lv[lvl].n;
  *n = (int)nl;
  if (!pc_u && !pc_v && !pc_idepth && !pc_color) return DSM_OK;
  if (!pc_u || !pc_v || !pc_idepth || !pc_color) return invalid("dsm_tracker_get_template: all four outputs or none");
  if (!nl) return DSM_OK;
  int rc = ensure_stage(ctx, 4 * nl);
  if (rc) return rc;
  float *su = ctx->d_stage, *sv = su + nl, *si = sv + nl, *sc = si + nl;
  launch_deinterleave_template(ctx->stream, (int)nl, t->d_pts[lvl], su, sv, si, sc);
  DSM_HIP(hipMemcpyAsync(pc_u, su, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipMemcpyAsync(pc_v, sv, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipMemcpyAsync(pc_idepth, si, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipMemcpyAsync(pc_color, sc, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

int dsm_tracker_upload_frame(dsm_tracker *t, int slot, const float *const *dIp, float ab_exposure) {
  if (!t || !dIp || slot < 0 || slot > 1) return invalid("dsm_tracker_upload_frame: bad argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  // The device keeps channel 0 only; channels 1 and 2 must be what FrameHessian::makeImages derives
  // from it -- which is all the reference ever passes -- and that is checked (dsm_params.frame_check, frame_grad_tol), not
  // assumed: per level the number of offending texels and the first of them.
  const size_t npx0 = (size_t)t->w * t->h;
  int rc = ensure_stage(ctx, 3 * npx0 + 4 * DSM_MAX_LEVELS);
  if (rc) return rc;
  int *d_bad = (int *)(ctx->d_stage + 3 * npx0); // [level]{count, first index}
  int h_bad[2 * DSM_MAX_LEVELS];
  for (int l = 0; l < DSM_MAX_LEVELS; l++) h_bad[2 * l] = 0, h_bad[2 * l + 1] = 0x7FFFFFFF;
  const bool check = t->params.frame_check != 0;
  if (check) DSM_HIP(hipMemcpyAsync(d_bad, h_bad, sizeof h_bad, hipMemcpyHostToDevice, ctx->stream));
  for (int l = 0; l < t->nlevels; l++) {
    const int wl = t->w >> l, hl = t->h >> l;
    DSM_HIP(hipMemcpyAsync(ctx->d_stage, dIp[l], (size_t)wl * hl * 12, hipMemcpyHostToDevice, ctx->stream));
    launch_dip_import(ctx->stream, wl, hl, ctx->d_stage, t->d_img[slot][l], check ? d_bad + 2 * l : nullptr, t->params.frame_grad_tol);
  }
  if (check) DSM_HIP(hipMemcpyAsync(h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  for (int l = 0; check && l < t->nlevels; l++)
    if (h_bad[2 * l]) {
      t->have_frame[slot] = false;
      const int wl = t->w >> l, idx = h_bad[2 * l + 1];
      char msg[320];
      snprintf(msg, sizeof msg,
               "dsm_tracker_upload_frame: level %d: %d texel(s) whose gradient channels are not the central differences of channel 0 "
               "(makeImages), the first at index %d (x = %d, y = %d); see dsm_params.frame_check / frame_grad_tol",
               l, h_bad[2 * l], idx, idx % wl, idx / wl);
      return invalid(msg);
    }
  t->desc.exposure[slot] = ab_exposure;
  t->have_frame[slot] = true;
  t->desc_dirty = true;
  return DSM_OK;
}

int dsm_tracker_upload_intensity(dsm_tracker *t, int slot, const float *const *I, float ab_exposure) {
  if (!t || !I || slot < 0 || slot > 1) return invalid("dsm_tracker_upload_intensity: bad argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  for (int l = 0; l < t->nlevels; l++) {
    if (!I[l]) return invalid("dsm_tracker_upload_intensity: null level");
    DSM_HIP(hipMemcpyAsync(t->d_img[slot][l], I[l], sizeof(float) * (size_t)(t->w >> l) * (t->h >> l), hipMemcpyHostToDevice, ctx->stream));
  }
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  t->desc.exposure[slot] = ab_exposure;
  t->have_frame[slot] = true;
  t->desc_dirty = true;
  return DSM_OK;
}

int dsm_tracker_upload_image(dsm_tracker *t, int slot, const float *image, float ab_exposure) {
  if (!t || !image || slot < 0 || slot > 1) return invalid("dsm_tracker_upload_image: bad argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  const size_t npx0 = (size_t)t->w * t->h;
  // the raw image is staged in a buffer of this tracker and slot, so that nothing but the host->device copy has to finish
  // before the call returns (the caller's buffer is free again); the pyramid kernels run on behind it.  With a pinned
  // caller buffer (dsm_host_alloc) the copy is a straight DMA.
  if (!t->d_raw[slot]) DSM_HIP(hipMalloc(&t->d_raw[slot], npx0 * sizeof(float)));
  DSM_HIP(hipMemcpyAsync(t->d_raw[slot], image, npx0 * 4, hipMemcpyHostToDevice, ctx->stream));
  DSM_HIP(hipEventRecord(ctx->copy_event, ctx->stream));
  launch_pyramid(ctx->stream, t->w, t->h, t->nlevels, t->d_raw[slot], t->d_img[slot]);
  DSM_HIP(hipGetLastError());
  DSM_HIP(hipEventSynchronize(ctx->copy_event));
  t->desc.exposure[slot] = ab_exposure;
  t->have_frame[slot] = true;
  t->desc_dirty = true;
  return DSM_OK;
}

// Target buffers of a hand-over: the frame slot itself (0, 1) or its back buffers (DSM_SLOT_NEXT_*), which
// dsm_frames_advance later swaps in.  Back buffers are allocated on first use.
static int upload_target(dsm_tracker *t, int slot, float **raw, float **img) {
  const size_t npx0 = (size_t)t->w * t->h;
  const int s = slot & 1;
  const bool back = slot >= 2;
  if (back) {
    for (int l = 0; l < t->nlevels; l++)
      if (!t->d_img_back[s][l]) {
        DSM_HIP(hipMalloc(&t->d_img_back[s][l], plane_bytes(t->w >> l, t->h >> l)));
        DSM_HIP(hipMemsetAsync(t->d_img_back[s][l], 0, plane_bytes(t->w >> l, t->h >> l), t->ctx->stream));
      }
    if (!t->d_raw_back[s]) DSM_HIP(hipMalloc(&t->d_raw_back[s], npx0 * sizeof(float)));
  } else if (!t->d_raw[s]) {
    DSM_HIP(hipMalloc(&t->d_raw[s], npx0 * sizeof(float)));
  }
  *raw = back ? t->d_raw_back[s] : t->d_raw[s];
  for (int l = 0; l < DSM_MAX_LEVELS; l++) img[l] = l < t->nlevels ? (back ? t->d_img_back[s][l] : t->d_img[s][l]) : nullptr;
  return DSM_OK;
}

static void upload_mark(dsm_tracker *t, int slot, float exposure) {
  if (slot >= 2) {
    t->have_back[slot & 1] = true;
    t->back_exposure[slot & 1] = exposure;
  } else {
    t->desc.exposure[slot] = exposure;
    t->have_frame[slot] = true;
    t->desc_dirty = true;
  }
}

// async = false: copies and pyramids ordered on the context's stream, returns when the copies are through.
// async = true: everything on the context's upload stream, returns at once; dsm_upload_wait waits for the copies.
static int upload_images_impl(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                              const float *ab_exposures, int pixel_type, size_t row_pitch_bytes, bool async, bool nowait = false) {
  if (!ctx || n < 0 || (n > 0 && (!trackers || !slots || !images)))
    return invalid("dsm_upload_images: bad argument");
  if (pixel_type != DSM_PIXEL_F32 && pixel_type != DSM_PIXEL_U8) return invalid("dsm_upload_images: bad pixel type");
  if (n == 0) return DSM_OK;
  DSM_HIP(hipSetDevice(ctx->device));
  const size_t px = pixel_type == DSM_PIXEL_U8 ? 1 : 4;
  const dsm_tracker *t0 = trackers[0];
  for (int i = 0; i < n; i++) {
    const dsm_tracker *t = trackers[i];
    if (!t || !images[i] || slots[i] < 0 || slots[i] > 3) return invalid("dsm_upload_images: bad entry");
    if (t->ctx != ctx) return invalid("dsm_upload_images: tracker of another context");
    if (t->w != t0->w || t->h != t0->h || t->nlevels != t0->nlevels)
      return invalid("dsm_upload_images: trackers of different geometry in one call");
    for (int j = 0; j < i; j++)
      if (trackers[j] == t && slots[j] == slots[i]) return invalid("dsm_upload_images: the same slot twice");
  }
  const size_t row = (size_t)t0->w * px;
  if (row_pitch_bytes != 0 && row_pitch_bytes < row) return invalid("dsm_upload_images: row pitch smaller than a row");
  if (!ctx->copy_stream) DSM_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (async && !ctx->upload_stream) {
    DSM_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    DSM_HIP(hipEventCreateWithFlags(&ctx->upload_copies_event, hipEventDisableTiming));
    DSM_HIP(hipEventCreateWithFlags(&ctx->upload_done_event, hipEventDisableTiming));
  }
  // the job table and the caller's buffers of the previous asynchronous hand-over are still being read until its
  // copies are through
  if (ctx->upload_pending) {
    DSM_HIP(hipEventSynchronize(ctx->upload_copies_event));
    ctx->upload_pending = false;
  }
  if (ctx->enqueue_pending) { // (dsm_upload_images_enqueue: its copy kernel still reads the job table and the caller's buffers)
    DSM_HIP(hipEventSynchronize(ctx->copy_event));
    ctx->enqueue_pending = false;
  }
  if (n > ctx->pyr_jobs_cap) {
    if (ctx->upload_stream) DSM_HIP(hipStreamSynchronize(ctx->upload_stream)); // pyramid kernels read the old table
    DSM_HIP(hipStreamSynchronize(ctx->stream));
    int rc = realloc_dev(&ctx->d_pyr_jobs, (size_t)n);
    if (rc) return rc;
    rc = realloc_pinned(&ctx->h_pyr_jobs, (size_t)n);
    if (rc) return rc;
    ctx->pyr_jobs_cap = n;
  }
  const size_t npx0 = (size_t)t0->w * t0->h;
  const size_t pitch = row_pitch_bytes ? row_pitch_bytes : row;
  // pinned caller buffers (dsm_host_alloc, hipHostMalloc, hipHostRegister) are read by the GPU directly
  bool all_pinned = true;
  uintptr_t align = (uintptr_t)row | (uintptr_t)pitch;
  for (int i = 0; i < n; i++) {
    float *raw = nullptr;
    int rc = upload_target(trackers[i], slots[i], &raw, ctx->h_pyr_jobs[i].img);
    if (rc) return rc;
    ctx->h_pyr_jobs[i].raw = raw;
    ctx->h_pyr_jobs[i].src = nullptr;
    if (all_pinned) {
      hipPointerAttribute_t attr{};
      if (hipPointerGetAttributes(&attr, images[i]) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
        ctx->h_pyr_jobs[i].src = attr.devicePointer;
        align |= (uintptr_t)attr.devicePointer;
      } else {
        (void)hipGetLastError(); // pageable memory: not an error
        all_pinned = false;
      }
    }
  }
  const hipStream_t work = async ? ctx->upload_stream : ctx->stream; // job table, copy kernel, pyramid kernels
  const bool u8 = pixel_type == DSM_PIXEL_U8;
  // the job table of the previous hand-over may still be read by its pyramid kernels on the other work stream
  if (async) {
    DSM_HIP(hipEventRecord(ctx->copy_event, ctx->stream));
    DSM_HIP(hipStreamWaitEvent(work, ctx->copy_event, 0));
  } else if (ctx->upload_stream) {
    DSM_HIP(hipStreamWaitEvent(work, ctx->upload_done_event, 0));
  }
  if (all_pinned) {
    const int unit = (align & 15) == 0 ? 16 : (align & 3) == 0 ? 4 : 1;
    DSM_HIP(hipMemcpyAsync(ctx->d_pyr_jobs, ctx->h_pyr_jobs, sizeof(dsm::PyrJob) * n, hipMemcpyHostToDevice, work));
    // asynchronous: few workgroups, so that the host reads (microseconds of latency each) do not sit in the memory
    // pipelines of the CUs the tracking kernels run on
    launch_host_rows_copy(work, ctx->d_pyr_jobs, n, (int)row, t0->h, pitch, unit, async ? ctx->async_copy_blocks : 1 << 20);
    DSM_HIP(hipGetLastError());
    DSM_HIP(hipEventRecord(async ? ctx->upload_copies_event : ctx->copy_event, work));
    launch_pyramid_batched(work, t0->w, t0->h, t0->nlevels, ctx->d_pyr_jobs, n, u8);
    DSM_HIP(hipGetLastError());
    if (!async && nowait)
      ctx->enqueue_pending = true; // (waited for by the next hand-over or dsm_upload_wait)
    else if (!async)
      DSM_HIP(hipEventSynchronize(ctx->copy_event)); // the caller's buffers are free; the pyramid kernels run on behind
  } else {
    // the staging buffers may still be read by the pyramid kernels of the previous hand-over
    DSM_HIP(hipEventRecord(ctx->copy_event, work));
    DSM_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_event, 0));
    DSM_HIP(hipMemcpyAsync(ctx->d_pyr_jobs, ctx->h_pyr_jobs, sizeof(dsm::PyrJob) * n, hipMemcpyHostToDevice, ctx->copy_stream));
    // groups: the pyramids of one group are built under the copies of the next
    const int group = 16;
    const int ngroups = (n + group - 1) / group;
    while ((int)ctx->upload_events.size() < ngroups) {
      hipEvent_t ev;
      DSM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      ctx->upload_events.push_back(ev);
    }
    for (int g = 0; g < ngroups; g++) {
      const int i0 = g * group, i1 = i0 + group < n ? i0 + group : n;
      for (int i = i0; i < i1; i++) {
        void *dst = const_cast<void *>(ctx->h_pyr_jobs[i].raw);
        if (pitch == row)
          DSM_HIP(hipMemcpyAsync(dst, images[i], npx0 * px, hipMemcpyHostToDevice, ctx->copy_stream));
        else
          DSM_HIP(hipMemcpy2DAsync(dst, row, images[i], pitch, row, (size_t)t0->h, hipMemcpyHostToDevice, ctx->copy_stream));
      }
      DSM_HIP(hipEventRecord(ctx->upload_events[g], ctx->copy_stream));
      DSM_HIP(hipStreamWaitEvent(work, ctx->upload_events[g], 0));
      launch_pyramid_batched(work, t0->w, t0->h, t0->nlevels, ctx->d_pyr_jobs + i0, i1 - i0, u8);
      DSM_HIP(hipGetLastError());
    }
    if (async)
      DSM_HIP(hipEventRecord(ctx->upload_copies_event, ctx->copy_stream));
    else
      DSM_HIP(hipStreamSynchronize(ctx->copy_stream)); // the caller's buffers are free; the pyramid kernels run on behind
  }
  if (async) {
    DSM_HIP(hipEventRecord(ctx->upload_done_event, work));
    ctx->upload_pending = true;
  }
  for (int i = 0; i < n; i++) upload_mark(trackers[i], slots[i], ab_exposures ? ab_exposures[i] : 1.0f);
  return DSM_OK;
}

int dsm_upload_images(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                      const float *ab_exposures, int pixel_type, size_t row_pitch_bytes) {
  return upload_images_impl(ctx, n, trackers, slots, images, ab_exposures, pixel_type, row_pitch_bytes, false);
}

int dsm_upload_images_async(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                            const float *ab_exposures, int pixel_type, size_t row_pitch_bytes) {
  return upload_images_impl(ctx, n, trackers, slots, images, ab_exposures, pixel_type, row_pitch_bytes, true);
}

int dsm_upload_images_enqueue(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots, const void *const *images,
                              const float *ab_exposures, int pixel_type, size_t row_pitch_bytes) {
  for (int i = 0; trackers && slots && i < n; i++)
    if (slots[i] > 1) return invalid("dsm_upload_images_enqueue: front buffers only (slots 0 / 1)");
  return upload_images_impl(ctx, n, trackers, slots, images, ab_exposures, pixel_type, row_pitch_bytes, false, true);
}

int dsm_upload_wait(dsm_context *ctx) {
  if (!ctx) return invalid("dsm_upload_wait: null context");
  if (ctx->upload_pending) {
    DSM_HIP(hipSetDevice(ctx->device));
    DSM_HIP(hipEventSynchronize(ctx->upload_copies_event));
    ctx->upload_pending = false;
  }
  if (ctx->enqueue_pending) {
    DSM_HIP(hipSetDevice(ctx->device));
    DSM_HIP(hipEventSynchronize(ctx->copy_event));
    ctx->enqueue_pending = false;
  }
  return DSM_OK;
}

int dsm_frames_advance(dsm_context *ctx, int n, dsm_tracker *const *trackers, const int *slots) {
  if (!ctx || n < 0 || (n > 0 && (!trackers || !slots))) return invalid("dsm_frames_advance: bad argument");
  for (int i = 0; i < n; i++) {
    const dsm_tracker *t = trackers[i];
    if (!t || t->ctx != ctx || slots[i] < 0 || slots[i] > 1) return invalid("dsm_frames_advance: bad entry");
    if (!t->have_back[slots[i]]) return invalid("dsm_frames_advance: nothing was handed over to DSM_SLOT_NEXT_* of this slot");
    for (int j = 0; j < i; j++)
      if (trackers[j] == t && slots[j] == slots[i]) return invalid("dsm_frames_advance: the same slot twice");
  }
  if (n == 0) return DSM_OK;
  DSM_HIP(hipSetDevice(ctx->device));
  // everything enqueued on the context's stream from here on sees the finished pyramids
  if (ctx->upload_stream) DSM_HIP(hipStreamWaitEvent(ctx->stream, ctx->upload_done_event, 0));
  for (int i = 0; i < n; i++) {
    dsm_tracker *t = trackers[i];
    const int s = slots[i];
    for (int l = 0; l < t->nlevels; l++) {
      float *f = t->d_img[s][l];
      t->d_img[s][l] = t->d_img_back[s][l];
      t->d_img_back[s][l] = f;
      t->desc.lv[l].img[s] = t->d_img[s][l];
    }
    float *r = t->d_raw[s];
    t->d_raw[s] = t->d_raw_back[s];
    t->d_raw_back[s] = r;
    t->desc.exposure[s] = t->back_exposure[s];
    t->have_frame[s] = true;
    t->have_back[s] = false;
    t->desc_dirty = true;
  }
  return DSM_OK;
}

int dsm_host_alloc(size_t bytes, void **out) {
  if (!out || bytes == 0) return invalid("dsm_host_alloc: bad argument");
  *out = nullptr;
  DSM_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return DSM_OK;
}

int dsm_host_free(void *p) {
  if (p) DSM_HIP(hipHostFree(p));
  return DSM_OK;
}

int dsm_tracker_get_frame(dsm_tracker *t, int slot, int lvl, float *dIp_out) {
  if (!t || !dIp_out || slot < 0 || slot > 1 || lvl < 0 || lvl >= t->nlevels) return invalid("dsm_tracker_get_frame: bad argument");
  dsm_context *ctx = t->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  const int wl = t->w >> lvl, hl = t->h >> lvl;
  const size_t npx = (size_t)wl * hl;
  int rc = ensure_stage(ctx, 3 * npx);
  if (rc) return rc;
  launch_dip_export(ctx->stream, wl, hl, t->d_img[slot][lvl], ctx->d_stage);
  DSM_HIP(hipMemcpyAsync(dIp_out, ctx->d_stage, npx * 12, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  return DSM_OK;
}

int dsm_tracker_ref_frame_id(dsm_tracker *t) { return t ? t->ref_frame_id : -1; }

int dsm_reduction_geometry(dsm_tracker *t, int lvl, int n, int *threads, int *pts_per_thread_out, int *chunks) {
  (void)lvl;
  if (!t) return invalid("dsm_reduction_geometry: null tracker");
  if (threads) *threads = kThreads;
  if (pts_per_thread_out) *pts_per_thread_out = pts_per_thread(n, t->desc.p.geometry);
  if (chunks) *chunks = num_chunks(n, t->desc.p.geometry);
  return DSM_OK;
}

// ---- common batch plumbing ------------------------------------------------------------------
} // extern "C"
namespace dsm {
int check_ready(dsm_tracker *t, int mode) {
  if (!t->have_k || !t->have_ref || !t->have_frame[mode == 1 ? 1 : 0]) {
    set_error(mode ? "optimize_scale needs make_k, set_ref and the right frame (slot 1)"
                   : "track needs make_k, set_ref and the new left frame (slot 0)");
    return DSM_ERR_STATE;
  }
  return DSM_OK;
}
} // namespace dsm
extern "C" {

// ts: n trackers of a call in mode `mode`, followed by n2 trackers of the companion segment in mode `mode2`
static int prepare_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, int mode, int n2 = 0, int mode2 = 1) {
  if (!ctx || n <= 0 || n2 < 0 || !ts) return invalid("batch: bad argument");
  DSM_HIP(hipSetDevice(ctx->device));
  int ps = 0;
  for (int i = 0; i < n + n2; i++) {
    dsm_tracker *t = ts[i];
    if (!t || t->ctx != ctx) return invalid("batch: tracker does not belong to this context");
    if (t->w != ts[0]->w || t->h != ts[0]->h || t->nlevels != ts[0]->nlevels)
      return invalid("batch: all trackers must share image size and levels");
    int rc = check_ready(t, i < n ? mode : mode2);
    if (rc) return rc;
    const int need = 2 * max_chunks_upto(t->w * t->h) * kPartialStride; // second half: the speculative candidate's partials
    if (need > ps) ps = need;
  }
  int rc = ensure_batch_capacity(ctx, n + n2, ps);
  if (rc) return rc;
  rc = sync_descs(ctx, ts, n + n2);
  if (rc) return rc;
  for (int i = 0; i < n + n2; i++) ctx->h_tracker_ptrs[i] = ts[i]->d_desc;
  DSM_HIP(hipMemcpyAsync(ctx->d_tracker_ptrs, ctx->h_tracker_ptrs, sizeof(TrackerDev *) * (n + n2), hipMemcpyHostToDevice, ctx->stream));
  return DSM_OK;
}

} // extern "C"
namespace dsm {
hipEvent_t get_event(dsm_context *ctx, size_t idx) {
  while (ctx->ev_pool.size() <= idx) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    ctx->ev_pool.push_back(ev);
  }
  return ctx->ev_pool[idx];
}

// streams of the segments of a launch schedule: `ng` stream groups (the context's stream + ng - 1 extra ones) and, on
// request, the companion stream
// Do kernels of streams a and b run at the same time?  (The runtime maps streams onto a few hardware queues -- four by default --
// round robin, together with every other stream of the process; two streams that share a queue serialise.  Measured in round 5: a
// hipMemset on the null stream in dsm_tracker_create shifted the assignment, two of the three stream groups of dsm_stream_* landed on
// one queue, and the bench lost 6 % on 512 frames, 15 % on 256 and 30 % on the sparse template -- profiles/r05_ab_bisect.log.)
// A kernel that stays resident for 400 us on a, an empty one on b behind it in host order: b's finishes early only on another queue.
static int streams_overlap(dsm_context *ctx, hipStream_t a, hipStream_t b, bool *overlap) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
  const double wait_ms = 0.4;
  hipEvent_t e0, e1;
  DSM_HIP(hipEventCreate(&e0));
  DSM_HIP(hipEventCreate(&e1));
  int rc = DSM_OK;
  float ms = 0.f;
  hipError_t e = hipSuccess;
  for (int pass = 0; pass < 2 && e == hipSuccess; pass++) { // (pass 0 loads the two kernels)
    e = hipEventRecord(e0, a);
    launch_queue_probe_wait(a, pass == 0 ? 1 : (long long)(wait_ms * khz));
    launch_queue_probe_empty(b);
    if (e == hipSuccess) e = hipEventRecord(e1, b);
    if (e == hipSuccess) e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
  }
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (e != hipSuccess) rc = hip_fail(e, "queue probe", __FILE__, __LINE__);
  *overlap = ms < 0.5 * wait_ms;
  return rc;
}
// a new stream whose kernels run concurrently with those of every stream in `with` (up to 12 candidates: the runtime hands out its
// queues round robin, so a few rejected candidates later one on a free queue comes up); none found -- fewer hardware queues than
// streams wanted (GPU_MAX_HW_QUEUES) --: the last candidate, counted in ctx->streams_sharing_a_queue
static int create_concurrent_stream(dsm_context *ctx, const std::vector<hipStream_t> &with, hipStream_t *out) {
  std::vector<hipStream_t> rejected;
  hipStream_t found = nullptr;
  int rc = DSM_OK;
  for (int attempt = 0; attempt < 12 && !found && rc == DSM_OK; attempt++) {
    hipStream_t st;
    const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) {
      rc = hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
      break;
    }
    bool ok = true;
    for (size_t i = 0; i < with.size() && ok && rc == DSM_OK; i++) rc = streams_overlap(ctx, with[i], st, &ok);
    if (ok && rc == DSM_OK)
      found = st;
    else
      rejected.push_back(st);
  }
  if (!found && rc == DSM_OK && !rejected.empty()) {
    found = rejected.back();
    rejected.pop_back();
    ctx->streams_sharing_a_queue++;
  }
  for (hipStream_t st : rejected) hipStreamDestroy(st);
  *out = found;
  return rc;
}

int ensure_streams(dsm_context *ctx, int ng, bool companion) {
  auto in_use = [&]() {
    std::vector<hipStream_t> v{ctx->stream};
    v.insert(v.end(), ctx->extra_streams.begin(), ctx->extra_streams.end());
    if (ctx->companion_stream) v.push_back(ctx->companion_stream);
    return v;
  };
  while ((int)ctx->extra_streams.size() < ng - 1) { // (the groups first: they carry the large kernels)
    hipStream_t st;
    const int rc = create_concurrent_stream(ctx, in_use(), &st);
    if (rc) return rc;
    ctx->extra_streams.push_back(st);
    hipEvent_t ev;
    DSM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ctx->join_events.push_back(ev);
  }
  if (companion && !ctx->companion_stream) {
    const int rc = create_concurrent_stream(ctx, in_use(), &ctx->companion_stream);
    if (rc) return rc;
    DSM_HIP(hipEventCreateWithFlags(&ctx->companion_event, hipEventDisableTiming));
  }
  return DSM_OK;
}

// timing enabled: per-level sums and interval unions of the eval dispatches bracketed by ev_pool[2 i], ev_pool[2 i + 1]
// (level ev_lvl[i]), relative to ev_total[0] (the stream groups' dispatches overlap; the union is the time during which the
// level's kernel ran at all)
void collect_eval_timing(dsm_context *ctx, const std::vector<int> &ev_lvl, int nlevels, dsm_stats &st) {
  std::vector<std::pair<float, float>> iv[DSM_MAX_LEVELS];
  for (size_t i = 0; i < ev_lvl.size(); i++) {
    float m = 0, a = 0;
    if (hipEventElapsedTime(&m, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]) == hipSuccess &&
        hipEventElapsedTime(&a, ctx->ev_total[0], ctx->ev_pool[2 * i]) == hipSuccess) {
      st.eval_kernel_ms[ev_lvl[i]] += m;
      st.eval_dispatches[ev_lvl[i]]++;
      iv[ev_lvl[i]].push_back(std::make_pair(a, a + m));
    }
  }
  for (int l = 0; l < nlevels; l++) {
    std::sort(iv[l].begin(), iv[l].end());
    double busy = 0, cs = 0, ce = -1;
    for (auto &p : iv[l]) {
      if (ce < 0) {
        cs = p.first, ce = p.second;
      } else if (p.first > ce) {
        busy += ce - cs;
        cs = p.first, ce = p.second;
      } else if (p.second > ce)
        ce = p.second;
    }
    if (ce >= 0) busy += ce - cs;
    st.eval_kernel_union_ms[l] += busy; // (+=: a stream's statistics are cumulative; the batch calls clear theirs per call)
  }
}
} // namespace dsm
extern "C" {

// runs the device LM state machine for a batch (mode 0 = trackNewestCoarse, 1 = optimizeScale, 2 = loop-closure pose).
// n2 > 0: problems [n, n + n2) form a COMPANION segment in mode2 (dsm_track_and_scale_batch: the keyframes' scale
// optimisations next to the frames' tracking).  The two segments are independent problems; the companion's launches go to a
// stream of their own, so its latency-bound chains of small launches run under the main segment's kernels instead of after
// them.  Launch-per-step form only.  Statistics of the companion: ctx->stats2.
static int run_lm_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, int mode, int coarsest, int n2 = 0, int mode2 = 1) {
  const int nlevels = ts[0]->nlevels;
  if (coarsest < 0 || coarsest >= nlevels) return invalid("coarsest level out of range"); // :457 / :856
  const dsm_params &P = ts[0]->params;
  const int N = n + n2;
  memset(&ctx->stats, 0, sizeof ctx->stats);
  memset(&ctx->stats2, 0, sizeof ctx->stats2);
  DSM_HIP(hipEventRecord(ctx->ev_total[0], ctx->stream));
  DSM_HIP(hipMemcpyAsync(ctx->d_start, ctx->h_start, sizeof(StartInfo) * N, hipMemcpyHostToDevice, ctx->stream));
  launch_lm(ctx->stream, mode, LM_OP_START, coarsest, n, ctx->d_tracker_ptrs, ctx->d_states, ctx->d_partials,
            ctx->partial_stride, ctx->d_start, nullptr, ctx->d_status);
  if (n2 > 0)
    launch_lm(ctx->stream, mode2, LM_OP_START, coarsest, n2, ctx->d_tracker_ptrs + n, ctx->d_states + n,
              ctx->d_partials + (size_t)n * ctx->partial_stride, ctx->partial_stride, ctx->d_start + n, nullptr, ctx->d_status + 2 * n);
  // Work-queue form: one launch of persistent workgroups for the whole call (queue_kernel).  Scheduling only:
  // results are bit-identical to the launch-per-step form below.
  // Default rule (measured): the queue form wins while the batch is large enough to fill the
  // persistent workgroups and small enough that its per-item cost (a few microseconds per chunk) does not add up --
  // up to about 24 k finest-level chunks per call (about 200 dense KITTI frames; thousands of sparse ones).
  bool use_queue = P.work_queue >= 2 && n2 == 0;
  if (P.work_queue == 1 && n >= 32 && n2 == 0) {
    long long chunks0 = 0;
    for (int i = 0; i < n; i++) chunks0 += level_chunks(ts[i]->desc, 0);
    use_queue = chunks0 <= 24576; // (256 dense S2 frames = 29.7 k chunks: 4 % slower than the launch form; 192: faster)
  }
  if (use_queue) {
    int max_items = 1;
    for (int L = 0; L <= coarsest; L++)
      for (int i = 0; i < n; i++) {
        const int c = level_chunks(ts[i]->desc, L);
        if (c > max_items) max_items = c;
      }
    if (max_items >= (1 << kQueueChunkBits) || n >= (1 << (32 - kQueueChunkBits))) return invalid("work queue: batch or level too large");
    size_t qcap = 1024; // outstanding items <= problems x chunks of one evaluation: a problem has one evaluation in flight
    while (qcap < (size_t)n * max_items) qcap <<= 1;
    if (qcap > ctx->qcap) {
      if (ctx->d_qitems) DSM_HIP(hipFree(ctx->d_qitems));
      ctx->d_qitems = nullptr;
      ctx->qcap = 0;
      DSM_HIP(hipMalloc(&ctx->d_qitems, qcap * sizeof(unsigned long long)));
      ctx->qcap = qcap;
    }
    if (!ctx->d_queue) DSM_HIP(hipMalloc(&ctx->d_queue, sizeof(WorkQueue)));
    if (!ctx->h_queue) DSM_HIP(hipHostMalloc(&ctx->h_queue, sizeof(WorkQueue), hipHostMallocDefault));
    DSM_HIP(hipMemsetAsync(ctx->d_queue, 0, sizeof(WorkQueue), ctx->stream));
    DSM_HIP(hipMemsetAsync(ctx->d_qitems, 0, ctx->qcap * sizeof(unsigned long long), ctx->stream)); // stale publication tags
    if (ctx->queue_blocks[mode] == 0) {
      hipDeviceProp_t prop;
      DSM_HIP(hipGetDeviceProperties(&prop, ctx->device));
      const int per_cu = queue_kernel_blocks_per_cu(mode);
      if (per_cu < 1) return invalid("work queue: kernel does not fit the device");
      ctx->queue_blocks[mode] = per_cu * prop.multiProcessorCount; // all workgroups must be co-resident
    }
    hipEvent_t qa = ctx->timing ? get_event(ctx, 0) : nullptr, qb = ctx->timing ? get_event(ctx, 1) : nullptr;
    if (qa) DSM_HIP(hipEventRecord(qa, ctx->stream));
    launch_queue(ctx->stream, mode, ctx->queue_blocks[mode], n, ctx->d_tracker_ptrs, ctx->d_states, ctx->d_partials,
                 ctx->partial_stride, ctx->d_tickets, ctx->d_queue, ctx->d_qitems, (unsigned)(ctx->qcap - 1));
    if (qb) DSM_HIP(hipEventRecord(qb, ctx->stream));
    DSM_HIP(hipGetLastError());
    DSM_HIP(hipMemcpyAsync(ctx->h_queue, ctx->d_queue, sizeof(WorkQueue), hipMemcpyDeviceToHost, ctx->stream));
    ctx->stats.queue_blocks = ctx->queue_blocks[mode];
  }
  size_t ev_used = 0;
  std::vector<int> ev_lvl;
  int rc_build = DSM_OK;
  // Launch schedule.  The LM loop is sequential per problem and its length is data dependent
  // (TrackerAndScaler.cpp:477,505,588,601); polling the device after every few launches costs a
  // host round trip each time.  Instead every level gets a speculative number of (eval, lm) launch
  // pairs -- the largest count any problem needed in recent calls plus slack -- and the state is
  // read back ONCE per pass.  Problems that finish a level early idle through the remaining
  // launches (their workgroups exit on the first instruction); problems that need more simply stay
  // at their level and are continued by the next pass.  Results do not depend on the schedule.
  int worst[DSM_MAX_LEVELS], grid_x[DSM_MAX_LEVELS], level_pts[DSM_MAX_LEVELS];
  bool spec[DSM_MAX_LEVELS];
  for (int L = 0; L < nlevels; L++) {
    int max_chunks = 1, max_it = 0, max_n = 0;
    for (int i = 0; i < N; i++) {
      const int c = level_chunks(ts[i]->desc, L);
      if (c > max_chunks) max_chunks = c;
      if (ts[i]->desc.lv[L].n > max_n) max_n = ts[i]->desc.lv[L].n;
      const int it_i = ts[i]->params.fixed_schedule > 0 ? ts[i]->params.fixed_schedule : ts[i]->params.max_iterations[L];
      if (it_i > max_it) max_it = it_i;
    }
    grid_x[L] = max_chunks < 8 ? max_chunks : round8(max_chunks);
    level_pts[L] = max_n;
    worst[L] = 2 * (7 + (max_it > 0 ? max_it : 0)); // upper bound of evaluations at one level
    // Speculative second candidate (dsm_device.hpp): doubles the evaluation work of a step to save the launches of
    // rejected steps.  It pays where a launch is latency- and not bandwidth-bound and rejections come in runs: the
    // small levels (a few thousand points).  Measured (S2 dense, launch form): 64 frames +12 %, 512 frames +-0 %, one
    // frame -1 % when applied to every level (the fine levels end on their first rejection), DESIGN.md section 4.3.
    // "Latency-bound" is a property of the launch, not of the level alone: a stream group's launch over G problems of n points
    // evaluates G * n points, and above about a million of them the doubled work costs more than the saved launches give back
    // (S2 dense, 512 + 103 problems in two groups, level 3 = 2.3 M points per launch: 51.3-51.8 k frames/s with the second
    // candidate there, 52.3-52.6 k without; levels 4 and 5, 0.58 M and 0.14 M points, make no measurable difference).
    {
      int groups = ctx->n_streams < 1 ? 1 : ctx->n_streams;
      if (groups > N) groups = N;
      const long long launch_points = (long long)((N + groups - 1) / groups) * max_n;
      spec[L] = P.fixed_schedule <= 0 && (P.speculate >= 2 || (P.speculate == 1 && max_n <= 8192 && launch_points <= 1000000ll));
    }
  }
  // the learnt "rounds most problems need" belong to calls of this size and form: another batch-size bucket (powers of four)
  // or form must not inherit them (a batch of hundreds following a single-frame call would switch to compact launches after
  // one round, covering almost every problem)
  {
    int bucket = 0;
    for (int v = N; v >= 4; v >>= 2) bucket++;
    const int key = bucket * 4 + (use_queue ? 2 : 0) + (P.persistent_coarse != 0 ? 1 : 0);
    if (key != ctx->sched_bulk_key) {
      for (int m = 0; m < 3; m++)
        for (int l = 0; l < DSM_MAX_LEVELS; l++) ctx->sched_bulk[m][l] = 1 << 30;
      ctx->sched_bulk_key = key;
    }
  }
  int *sched = ctx->sched[mode], *sched2 = ctx->sched[mode2];
  const int *bulk = ctx->sched_bulk[mode], *bulk2 = ctx->sched_bulk[mode2];
  int ng = ctx->n_streams < 1 ? 1 : ctx->n_streams;
  if (ng > n) ng = n;
  {
    const int rc = ensure_streams(ctx, ng, n2 > 0);
    if (rc) return rc;
  }
  int top = coarsest, top2 = n2 > 0 ? coarsest : -1;
  if (!use_queue && P.persistent_coarse < 0) {
    // Small levels as a CHAIN (chain_kernel): every problem whose evaluation at the coarsest level is one chunk runs its LM loop in one
    // launch, down to its first level of several chunks.  One launch per stream group and one for the companion segment.
    auto one_chunk = [&](int i, int L) { return level_chunks(ts[i]->desc, L) <= 1; };
    bool any = false;
    for (int i = 0; i < N && !any; i++) any = one_chunk(i, coarsest);
    if (any) {
      if (ng > 1 || top2 >= 0) {
        DSM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
        for (int g = 1; g < ng; g++) DSM_HIP(hipStreamWaitEvent(ctx->extra_streams[g - 1], ctx->fork_event, 0));
        if (top2 >= 0) DSM_HIP(hipStreamWaitEvent(ctx->companion_stream, ctx->fork_event, 0));
      }
      for (int g = 0; g < ng; g++) {
        const int g0 = (int)((long long)n * g / ng), g1 = (int)((long long)n * (g + 1) / ng);
        if (g1 <= g0) continue;
        launch_chain(g == 0 ? ctx->stream : ctx->extra_streams[g - 1], mode, g1 - g0, ctx->d_tracker_ptrs + g0, ctx->d_states + g0, ctx->d_status + 2 * g0);
      }
      if (n2 > 0) launch_chain(ctx->companion_stream, mode2, n2, ctx->d_tracker_ptrs + n, ctx->d_states + n, ctx->d_status + 2 * n);
      ctx->stats.coarse_launches = 1;
      auto first_launch_level = [&](int i0, int i1) { // (the launch-per-step schedule starts at the first level some problem cannot chain through)
        int t = -1;
        for (int i = i0; i < i1; i++)
          for (int L = coarsest; L > t; L--)
            if (!one_chunk(i, L)) {
              t = L;
              break;
            }
        return t;
      };
      top = first_launch_level(0, n);
      if (n2 > 0) top2 = first_launch_level(n, N);
    }
  }
  if (!use_queue && P.persistent_coarse > 0) {
    // Small levels: the whole LM loop in one launch per problem on LDS-resident data (coarse_kernel), one launch per stream
    // group and one for the companion segment.  A problem is handed back (still RUNNING) at the first level whose target
    // plane does not fit the kernel's LDS arena.
    const int max_px = P.persistent_coarse;
    bool any = false;
    for (int i = 0; i < N && !any; i++) any = coarse_level_fits(ts[i]->w >> coarsest, ts[i]->h >> coarsest, ts[i]->desc.lv[coarsest].n, ts[i]->desc.p.geometry, max_px);
    if (any) {
      if (ng > 1 || top2 >= 0) {
        DSM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
        for (int g = 1; g < ng; g++) DSM_HIP(hipStreamWaitEvent(ctx->extra_streams[g - 1], ctx->fork_event, 0));
        if (top2 >= 0) DSM_HIP(hipStreamWaitEvent(ctx->companion_stream, ctx->fork_event, 0));
      }
      const bool cspec = P.speculate >= 1 && P.fixed_schedule <= 0;
      for (int g = 0; g < ng; g++) {
        const int g0 = (int)((long long)n * g / ng), g1 = (int)((long long)n * (g + 1) / ng);
        if (g1 <= g0) continue;
        launch_coarse(g == 0 ? ctx->stream : ctx->extra_streams[g - 1], mode, g1 - g0, ctx->d_tracker_ptrs + g0, ctx->d_states + g0,
                      ctx->d_status + 2 * g0, max_px, cspec);
      }
      if (n2 > 0)
        launch_coarse(ctx->companion_stream, mode2, n2, ctx->d_tracker_ptrs + n, ctx->d_states + n, ctx->d_status + 2 * n, max_px, cspec);
      ctx->stats.coarse_launches = 1;
      // The launch-per-step schedule starts at the first level some problem cannot run there.  No join: each stream's next
      // launches are ordered behind its own coarse launch (the pass loop's fork adds the main stream's only).
      auto first_launch_level = [&](int i0, int i1) {
        int t = -1;
        for (int i = i0; i < i1; i++)
          for (int L = coarsest; L > t; L--)
            if (!coarse_level_fits(ts[i]->w >> L, ts[i]->h >> L, ts[i]->desc.lv[L].n, ts[i]->desc.p.geometry, max_px)) {
              t = L;
              break;
            }
        return t;
      };
      top = first_launch_level(0, n);
      if (n2 > 0) top2 = first_launch_level(n, N);
    }
  }
  // Segments of the launch schedule: the stream groups of the main batch (the batch is split into `ng` contiguous groups,
  // each with its own HIP stream: a group's lm_kernel -- one small workgroup per problem -- and its small-level eval kernels
  // leave most of the chip idle; another group's kernels fill it) and the companion segment.  Per-problem results do not
  // depend on the split.
  struct Seg {
    hipStream_t st;
    int i0, i1, mode;
    bool companion;
    int rows; // >= 0: compact launches over the first `rows` entries of the segment's row map; -1: one row per problem
  };
  std::vector<Seg> segs;
  for (int g = 0; g < ng; g++) {
    const int g0 = (int)((long long)n * g / ng), g1 = (int)((long long)n * (g + 1) / ng);
    if (g1 > g0) segs.push_back(Seg{g == 0 ? ctx->stream : ctx->extra_streams[g - 1], g0, g1, mode, false, -1});
  }
  if (n2 > 0) segs.push_back(Seg{ctx->companion_stream, n, N, mode2, true, -1});
  // one (evaluate, step) round of a segment at level L
  auto launch_round = [&](const Seg &sg, int L, int k, int pass) -> int {
    const int np = sg.i1 - sg.i0, rows = sg.rows >= 0 ? sg.rows : np;
    if (rows == 0) return DSM_OK;
    const int *rowmap = sg.rows >= 0 ? ctx->d_rowmap + sg.i0 : nullptr;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (ctx->timing && !sg.companion) {
      ea = get_event(ctx, ev_used++);
      eb = get_event(ctx, ev_used++);
      ev_lvl.push_back(L);
      if (ea) DSM_HIP(hipEventRecord(ea, sg.st));
    }
    // levels >= 1: the eval kernel's last-arriving workgroup per problem can perform the LM step itself (one launch per
    // round instead of two): for launches of few problems -- small batches (measured: -6 % latency for one frame in flight,
    // -11 % throughput at 256) and compact launches over a handful of stragglers.
    // (Measured on 512 all-distinct S2 frames, same box: fused compact rounds 39.7-40.2 k frames/s, pairs above 8 rows
    // 38.6-38.8 k, the speculative candidate on every level of the compact rounds 38.9 k: the pair's second launch and the
    // doubled rows cost what they save.)
    const bool fused = L > 0 && (P.fuse_lm >= 2 || (P.fuse_lm == 1 && (np <= 8 || sg.rows >= 0)));
    // Large levels: the residual-only evaluations (the level's last ones, tracker_kernels.hip) get a launch of their own
    // behind the full ones -- an instantiation without the 45 accumulators, 30-41 VGPRs = eight waves per SIMD instead
    // of four or five.  Never in a level's first round (its evaluation is the level's first).  Measured (S2 dense, 512
    // frames): level-0 evaluations 4.62 -> 4.40 ms per step, 54.3 -> 55.7 k frames/s with levels 0 and 1 split; with
    // level 2 as well 53.7-55.3 k (the extra launch costs more than it gives there); one frame in flight 0.70 -> 0.74 ms
    // (three more launches), hence the floor on the points per launch.
    const bool split_ro = !sg.companion && !fused && k > 0 && level_pts[L] >= 100000 && (long long)rows * level_pts[L] >= 8000000ll;
    launch_eval(sg.st, sg.mode, L, grid_x[L], rows, ctx->d_tracker_ptrs + sg.i0, ctx->d_states + sg.i0,
                ctx->d_partials + (size_t)sg.i0 * ctx->partial_stride, ctx->partial_stride, fused ? ctx->d_tickets + sg.i0 : nullptr,
                ctx->d_status + 2 * sg.i0, spec[L], split_ro, rowmap);
    if (eb) DSM_HIP(hipEventRecord(eb, sg.st));
    if (!fused)
      launch_lm(sg.st, sg.mode, LM_OP_STEP, L, rows, ctx->d_tracker_ptrs + sg.i0, ctx->d_states + sg.i0,
                ctx->d_partials + (size_t)sg.i0 * ctx->partial_stride, ctx->partial_stride, nullptr, nullptr, ctx->d_status + 2 * sg.i0,
                spec[L], rowmap);
    return DSM_OK;
  };
  // row map of a segment: the problems (relative to its first) for which `keep(status, level)` holds, from h_status
  auto build_rowmap = [&](Seg &sg, auto keep) -> int {
    int r = 0;
    for (int i = sg.i0; i < sg.i1; i++)
      if (keep(ctx->h_status[2 * i], ctx->h_status[2 * i + 1])) ctx->h_rowmap[sg.i0 + r++] = i - sg.i0;
    sg.rows = r;
    if (r > 0) DSM_HIP(hipMemcpyAsync(ctx->d_rowmap + sg.i0, ctx->h_rowmap + sg.i0, sizeof(int) * r, hipMemcpyHostToDevice, sg.st));
    return DSM_OK;
  };
  for (int pass = 0; !use_queue; pass++) {
    if (ng > 1 || top2 >= 0) { // fork: the extra streams start after everything enqueued on the main stream so far
      DSM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
      for (int g = 1; g < ng; g++) DSM_HIP(hipStreamWaitEvent(ctx->extra_streams[g - 1], ctx->fork_event, 0));
      if (top2 >= 0) DSM_HIP(hipStreamWaitEvent(ctx->companion_stream, ctx->fork_event, 0));
    }
    // Later passes only continue what the previous one left unfinished: compact launches over the problems still running
    // (a launch over all problems of a batch costs ~1.6 ns per idle workgroup: 100 us for a level-0 grid of 512 problems).
    const bool compact_ok = P.adaptive_schedule != 0 && P.compact_tail != 0;
    for (Seg &sg : segs) {
      sg.rows = -1;
      if (pass > 0 && compact_ok) {
        rc_build = build_rowmap(sg, [](int st, int) { return st == ST_RUNNING; });
        if (rc_build) return rc_build;
      }
    }
    for (int L = top > top2 ? top : top2; L >= 0; L--) {
      int steps = P.adaptive_schedule ? sched[L] << (pass > 3 ? 3 : pass) : worst[L];
      if (steps > worst[L]) steps = worst[L];
      if (steps < 1) steps = 1;
      if (L > top) steps = 0;
      int steps2 = 0;
      if (L <= top2) {
        steps2 = P.adaptive_schedule ? sched2[L] << (pass > 3 ? 3 : pass) : worst[L];
        if (steps2 > worst[L]) steps2 = worst[L];
        if (steps2 < 1) steps2 = 1;
      }
      // Phase A: the rounds most problems need (up to the third quartile of what recent calls' problems took), one row per
      // problem.  Phase B, first pass only: ONE read-back of the level's status, then the remaining rounds as compact
      // launches over the problems still at this level -- the stragglers (a level's last rounds are needed by a handful of
      // problems: 52 launches at levels 2 and 3 of 512 distinct S2 frames where the median problem needs 6 and 10).
      auto seg_steps = [&](const Seg &sg) { return sg.companion ? steps2 : steps; };
      auto seg_bulk = [&](const Seg &sg) {
        const int st = seg_steps(sg), bk = (sg.companion ? bulk2 : bulk)[L];
        const bool worth = compact_ok && pass == 0 && sg.i1 - sg.i0 >= 32 && (long long)grid_x[L] * (sg.i1 - sg.i0) >= 1024 && st - bk >= 3;
        return worth ? bk : st;
      };
      int kmax = 0;
      for (const Seg &sg : segs) kmax = seg_bulk(sg) > kmax ? seg_bulk(sg) : kmax;
      for (int k = 0; k < kmax; k++)
        for (int si = (int)segs.size() - 1; si >= 0; si--) { // (the companion's round first, as before)
          const Seg &sg = segs[si];
          if (k < seg_bulk(sg)) {
            const int rc = launch_round(sg, L, k, pass);
            if (rc) return rc;
          }
        }
      bool any_tail = false;
      for (const Seg &sg : segs) any_tail = any_tail || seg_bulk(sg) < seg_steps(sg);
      if (any_tail) {
        for (const Seg &sg : segs)
          if (seg_bulk(sg) < seg_steps(sg))
            DSM_HIP(hipMemcpyAsync(ctx->h_status + 2 * sg.i0, ctx->d_status + 2 * sg.i0, sizeof(int) * 2 * (sg.i1 - sg.i0), hipMemcpyDeviceToHost, sg.st));
        // the segments in the order their read-backs complete (the others keep their queues busy meanwhile): whichever stream is
        // idle already, else the first one still pending
        std::vector<char> tail_done(segs.size(), 0);
        DSM_HIP(hipGetLastError()); // (launch errors so far; a "not ready" answer below is consumed where it is returned)
        auto idle = [](hipStream_t st) {
          const hipError_t q = hipStreamQuery(st);
          if (q == hipErrorNotReady) (void)hipGetLastError();
          return q == hipSuccess;
        };
        for (size_t done = 0; done < segs.size(); done++) {
          int pick = -1;
          for (size_t si = 0; si < segs.size() && pick < 0; si++)
            if (!tail_done[si] && (seg_bulk(segs[si]) >= seg_steps(segs[si]) || idle(segs[si].st))) pick = (int)si;
          for (size_t si = 0; si < segs.size() && pick < 0; si++)
            if (!tail_done[si]) pick = (int)si;
          tail_done[pick] = 1;
          Seg &sg = segs[pick];
          const int kb = seg_bulk(sg), ks = seg_steps(sg);
          if (kb >= ks) continue;
          DSM_HIP(hipStreamSynchronize(sg.st));
          rc_build = build_rowmap(sg, [L](int st, int lv) { return st == ST_RUNNING && lv == L; });
          if (rc_build) return rc_build;
          for (int k = kb; k < ks; k++) {
            const int rc = launch_round(sg, L, k, pass);
            if (rc) return rc;
          }
          sg.rows = -1; // the next level starts with one row per problem again
        }
        ctx->stats.polls++;
      }
      ctx->stats.launches[L] += steps;
      ctx->stats2.launches[L] += steps2;
    }
    DSM_HIP(hipGetLastError()); // launch-configuration errors of the kernels enqueued above
    for (int g = 1; g < ng; g++) { // join
      DSM_HIP(hipEventRecord(ctx->join_events[g - 1], ctx->extra_streams[g - 1]));
      DSM_HIP(hipStreamWaitEvent(ctx->stream, ctx->join_events[g - 1], 0));
    }
    if (top2 >= 0) {
      DSM_HIP(hipEventRecord(ctx->companion_event, ctx->companion_stream));
      DSM_HIP(hipStreamWaitEvent(ctx->stream, ctx->companion_event, 0));
    }
    DSM_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int) * 2 * N, hipMemcpyDeviceToHost, ctx->stream));
    DSM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stats.polls++;
    top = top2 = -1;
    for (int i = 0; i < n; i++)
      if (ctx->h_status[2 * i] == ST_RUNNING && ctx->h_status[2 * i + 1] > top) top = ctx->h_status[2 * i + 1];
    for (int i = n; i < N; i++)
      if (ctx->h_status[2 * i] == ST_RUNNING && ctx->h_status[2 * i + 1] > top2) top2 = ctx->h_status[2 * i + 1];
    if (top < 0 && top2 < 0) break;
    if (pass > 64) {
      set_error("internal: LM state machine did not terminate within the launch bound");
      return DSM_ERR_STATE;
    }
  }
  DSM_HIP(hipMemcpyAsync(ctx->h_states, ctx->d_states, sizeof(LMState) * N, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipEventRecord(ctx->ev_total[1], ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  if (use_queue) {
    if (ctx->h_queue->error) {
      // leave the arrival counters clean for the next call
      DSM_HIP(hipMemsetAsync(ctx->d_tickets, 0, sizeof(int) * ctx->cap_prob, ctx->stream));
      DSM_HIP(hipStreamSynchronize(ctx->stream));
      set_error("internal: work-queue kernel gave up waiting (bounded wait expired)");
      return DSM_ERR_STATE;
    }
    ctx->stats.queue_items = ctx->h_queue->tail;
    if (ctx->timing && ctx->ev_pool.size() >= 2) {
      float qm = 0;
      if (hipEventElapsedTime(&qm, ctx->ev_pool[0], ctx->ev_pool[1]) == hipSuccess) ctx->stats.queue_kernel_ms = qm;
    }
  }
  float ms = 0;
  DSM_HIP(hipEventElapsedTime(&ms, ctx->ev_total[0], ctx->ev_total[1]));
  ctx->stats.total_ms = ms;
  if (ctx->timing) collect_eval_timing(ctx, ev_lvl, nlevels, ctx->stats);
  int need[DSM_MAX_LEVELS] = {0}, need2[DSM_MAX_LEVELS] = {0};
  ctx->stats2.total_ms = ms;
  for (int i = 0; i < N; i++) {
    const LMState &S = ctx->h_states[i];
    if (S.status == ST_RUNNING) {
      set_error("internal: LM state machine did not terminate within the launch bound");
      return DSM_ERR_STATE;
    }
    dsm_stats &st = i < n ? ctx->stats : ctx->stats2;
    int *nd = i < n ? need : need2;
    for (int l = 0; l < nlevels; l++) {
      if ((int)S.rounds[l] > nd[l]) nd[l] = (int)S.rounds[l]; // launches this problem needed at the level
      st.evals[l] += S.evals[l];
      st.evals_residual_only[l] += S.evals_ro[l];
      // compulsory bytes of one evaluation (what calcRes* reads): the template once + the target image once,
      // or, for a sparse template, the four 12-byte taps of every point if that is less
      const long long nl = ts[i]->desc.lv[l].n, img = 12ll * (ts[i]->w >> l) * (ts[i]->h >> l);
      st.algorithmic_bytes += S.evals[l] * (16ll * nl + (48ll * nl < img ? 48ll * nl : img));
    }
  }
  // the third quartile of the rounds per problem and level: where the next call's launches turn to the fused form
  {
    std::vector<int> r;
    for (int seg = 0; seg < (n2 > 0 ? 2 : 1); seg++) {
      const int i0 = seg ? n : 0, i1 = seg ? N : n;
      int *dst = ctx->sched_bulk[seg ? mode2 : mode];
      for (int l = 0; l < nlevels; l++) {
        r.clear();
        for (int i = i0; i < i1; i++) r.push_back((int)ctx->h_states[i].rounds[l]);
        std::nth_element(r.begin(), r.begin() + (3 * r.size()) / 4, r.end());
        dst[l] = r[(3 * r.size()) / 4] + 1;
      }
    }
  }
  // next call's schedule: what this batch needed plus one, decaying slowly towards it
  for (int l = 0; l < nlevels; l++) {
    const int want = need[l] + 1;
    const int decayed = sched[l] - (sched[l] + 7) / 8;
    sched[l] = want > decayed ? want : decayed;
    if (n2 > 0) {
      const int want2 = need2[l] + 1, decayed2 = sched2[l] - (sched2[l] + 7) / 8;
      sched2[l] = want2 > decayed2 ? want2 : decayed2;
    }
  }
  return DSM_OK;
}

int dsm_track_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, double *pose_io, double *aff_io,
                    int coarsest_lvl, const double *min_res_for_abort, double *last_residuals, double *flow_out,
                    int *good) {
  if (!pose_io || !aff_io) return invalid("dsm_track_batch: null pose/aff");
  int rc = prepare_batch(ctx, n, ts, 0);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    StartInfo &I = ctx->h_start[i];
    memset(&I, 0, sizeof I);
    memcpy(I.pose, pose_io + 7 * i, sizeof I.pose);
    memcpy(I.aff, aff_io + 2 * i, sizeof I.aff);
    for (int l = 0; l < DSM_MAX_LEVELS; l++)
      I.min_res[l] = min_res_for_abort ? min_res_for_abort[DSM_MAX_LEVELS * i + l] : std::numeric_limits<double>::quiet_NaN();
    I.scale = 1.0f;
    I.coarsest = coarsest_lvl;
  }
  rc = run_lm_batch(ctx, n, ts, 0, coarsest_lvl);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    const LMState &S = ctx->h_states[i];
    const bool ok = S.status == ST_GOOD;
    // the reference writes lastToNew_out / aff_g2l_out at :612-613, i.e. also when the later
    // affine plausibility checks (:615-626) fail, but not when a level aborts (:598)
    if (S.status == ST_GOOD || S.status == ST_BAD_AFFINE) {
      memcpy(pose_io + 7 * i, S.cur, sizeof(double) * 7);
      memcpy(aff_io + 2 * i, S.aff_cur, sizeof(double) * 2);
    }
    if (last_residuals) memcpy(last_residuals + DSM_MAX_LEVELS * i, S.last_residuals, sizeof(double) * DSM_MAX_LEVELS);
    if (flow_out) memcpy(flow_out + 3 * i, S.flow, sizeof(double) * 3);
    if (good) good[i] = ok ? 1 : 0;
  }
  return DSM_OK;
}

int dsm_tracker_track(dsm_tracker *t, double pose_io[7], double aff_io[2], int coarsest_lvl,
                      const double *min_res_for_abort, double *last_residuals, double flow_out[3], int *good) {
  if (!t) return invalid("null tracker");
  dsm_tracker *ts[1] = {t};
  return dsm_track_batch(t->ctx, 1, ts, pose_io, aff_io, coarsest_lvl, min_res_for_abort, last_residuals, flow_out, good);
}

int dsm_optimize_scale_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, float *scale_io, int coarsest_lvl,
                             float *err_out) {
  if (!scale_io) return invalid("dsm_optimize_scale_batch: null scale");
  int rc = prepare_batch(ctx, n, ts, 1);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    StartInfo &I = ctx->h_start[i];
    memset(&I, 0, sizeof I);
    I.pose[3] = 1.0;
    for (int l = 0; l < DSM_MAX_LEVELS; l++) I.min_res[l] = std::numeric_limits<double>::quiet_NaN();
    I.scale = scale_io[i];
    I.coarsest = coarsest_lvl;
  }
  rc = run_lm_batch(ctx, n, ts, 1, coarsest_lvl);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    const LMState &S = ctx->h_states[i];
    scale_io[i] = S.scale_cur;                               // :954
    if (err_out) err_out[i] = (float)S.last_residuals[0];    // :963
  }
  return DSM_OK;
}

// One step of many sequences: the frames' trackNewestCoarse AND the keyframes' optimizeScale in one call.  The scale problems
// are independent of the track problems (they read the keyframe tracker's template and its right frame, FrontEnd.cpp:992-998;
// the track problems read the same kind of template and the left frame), so their launches run on a stream of their own
// under the tracking kernels.  Same results as dsm_track_batch followed by dsm_optimize_scale_batch, bit for bit.
int dsm_track_and_scale_batch(dsm_context *ctx, int n, dsm_tracker *const *ts, double *pose_io, double *aff_io, int coarsest_lvl,
                              const double *min_res_for_abort, double *last_residuals, double *flow_out, int *good, int n_scale,
                              dsm_tracker *const *ts_scale, float *scale_io, float *err_out) {
  if (!pose_io || !aff_io || n < 1) return invalid("dsm_track_and_scale_batch: null pose/aff");
  if (n_scale < 0 || (n_scale && (!ts_scale || !scale_io))) return invalid("dsm_track_and_scale_batch: bad scale arguments");
  std::vector<dsm_tracker *> all(ts, ts + n);
  all.insert(all.end(), ts_scale, ts_scale + n_scale);
  int rc = prepare_batch(ctx, n, all.data(), 0, n_scale, 1);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    StartInfo &I = ctx->h_start[i];
    memset(&I, 0, sizeof I);
    memcpy(I.pose, pose_io + 7 * i, sizeof I.pose);
    memcpy(I.aff, aff_io + 2 * i, sizeof I.aff);
    for (int l = 0; l < DSM_MAX_LEVELS; l++)
      I.min_res[l] = min_res_for_abort ? min_res_for_abort[DSM_MAX_LEVELS * i + l] : std::numeric_limits<double>::quiet_NaN();
    I.scale = 1.0f;
    I.coarsest = coarsest_lvl;
  }
  for (int i = 0; i < n_scale; i++) {
    StartInfo &I = ctx->h_start[n + i];
    memset(&I, 0, sizeof I);
    I.pose[3] = 1.0;
    for (int l = 0; l < DSM_MAX_LEVELS; l++) I.min_res[l] = std::numeric_limits<double>::quiet_NaN();
    I.scale = scale_io[i];
    I.coarsest = coarsest_lvl;
  }
  rc = run_lm_batch(ctx, n, all.data(), 0, coarsest_lvl, n_scale, 1);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    const LMState &S = ctx->h_states[i];
    if (S.status == ST_GOOD || S.status == ST_BAD_AFFINE) { // as dsm_track_batch
      memcpy(pose_io + 7 * i, S.cur, sizeof(double) * 7);
      memcpy(aff_io + 2 * i, S.aff_cur, sizeof(double) * 2);
    }
    if (last_residuals) memcpy(last_residuals + DSM_MAX_LEVELS * i, S.last_residuals, sizeof(double) * DSM_MAX_LEVELS);
    if (flow_out) memcpy(flow_out + 3 * i, S.flow, sizeof(double) * 3);
    if (good) good[i] = S.status == ST_GOOD ? 1 : 0;
  }
  for (int i = 0; i < n_scale; i++) {
    const LMState &S = ctx->h_states[n + i];
    scale_io[i] = S.scale_cur;                             // :954
    if (err_out) err_out[i] = (float)S.last_residuals[0];  // :963
  }
  return DSM_OK;
}

int dsm_context_get_stats2(dsm_context *ctx, dsm_stats *out) {
  if (!ctx || !out) return invalid("dsm_context_get_stats2: null argument");
  *out = ctx->stats2;
  return DSM_OK;
}

// The untrapped branch of FrontEnd::optimizeScale (FrontEnd.cpp:995-1003): optimizeScale from every initial guess, keep the
// smallest positive error (the first one on ties).  The guesses are independent problems on the same template and right
// image: ONE batched call instead of eight sequential ones.
int dsm_tracker_optimize_scale_guesses(dsm_tracker *t, int n_guesses, const float *scale_guesses, int coarsest_lvl, float *scale_out,
                                       float *err_out, float *scales_all, float *errs_all) {
  if (!t || n_guesses < 1 || !scale_guesses || !scale_out || !err_out) return invalid("dsm_tracker_optimize_scale_guesses: bad argument");
  std::vector<dsm_tracker *> ts((size_t)n_guesses, t);
  std::vector<float> sc(scale_guesses, scale_guesses + n_guesses), er((size_t)n_guesses, 0.f);
  const int rc = dsm_optimize_scale_batch(t->ctx, n_guesses, ts.data(), sc.data(), coarsest_lvl, er.data());
  if (rc) return rc;
  float new_scale = 1.0f, scale_error = -1.0f; // :991-992
  for (int i = 0; i < n_guesses; i++) {
    const float cur_error = er[i];
    if (cur_error > 0 && (scale_error < 0 || scale_error > cur_error)) { // :998-1001 (NaN compares false: rejected)
      scale_error = cur_error;
      new_scale = sc[i];
    }
    if (scales_all) scales_all[i] = sc[i];
    if (errs_all) errs_all[i] = er[i];
  }
  *scale_out = new_scale;
  *err_out = scale_error;
  return DSM_OK;
}

int dsm_tracker_optimize_scale(dsm_tracker *t, float *scale_io, int coarsest_lvl, float *err_out) {
  if (!t) return invalid("null tracker");
  dsm_tracker *ts[1] = {t};
  return dsm_optimize_scale_batch(t->ctx, 1, ts, scale_io, coarsest_lvl, err_out);
}

// ---- loop-closure pose estimation (row N2): PoseEstimator::estimate, PoseEstimator.cpp:298-506 ----
} // extern "C"

struct dsm_pose_estimator {
  dsm_tracker *t = nullptr;
  std::vector<float> x, y, z;
};

extern "C" {

int dsm_pose_estimator_create(dsm_context *ctx, int w, int h, int nlevels, const dsm_params *params,
                              dsm_pose_estimator **out) {
  if (!out) return invalid("dsm_pose_estimator_create: out is NULL");
  const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float k1[4] = {1, 1, 0, 0};
  dsm_tracker *t = nullptr;
  int rc = dsm_tracker_create(ctx, w, h, nlevels, I4, k1, params, &t);
  if (rc) return rc;
  // the loop-closure point set is the same on every level: give every level the level-0 capacity
  for (int l = 1; l < nlevels; l++) {
    hipFree(t->d_pts[l]);
    t->d_pts[l] = nullptr;
    hipError_t e = hipMalloc(&t->d_pts[l], sizeof(float4) * ((size_t)w * h + kTemplatePad));
    if (e == hipSuccess) e = hipMemsetAsync(t->d_pts[l], 0, sizeof(float4) * ((size_t)w * h + kTemplatePad), ctx->stream);
    if (e != hipSuccess) {
      dsm_tracker_destroy(t);
      DSM_HIP(e);
    }
    t->pts_cap[l] = w * h;
    t->desc.lv[l].pts = t->d_pts[l];
  }
  t->desc_dirty = true;
  {
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      dsm_tracker_destroy(t);
      DSM_HIP(e);
    }
  }
  dsm_pose_estimator *pe = new dsm_pose_estimator();
  pe->t = t;
  *out = pe;
  return DSM_OK;
}

int dsm_pose_estimator_destroy(dsm_pose_estimator *pe) {
  if (!pe) return DSM_OK;
  dsm_tracker_destroy(pe->t);
  delete pe;
  return DSM_OK;
}

int dsm_pose_estimator_estimate(dsm_pose_estimator *pe, int n_pts, const double *xyz, const float *const *ref_colors,
                                float ref_ab_exposure, const float *const *new_dIp, float new_ab_exposure,
                                const float new_cam[4], int coarsest_lvl, double ref_to_new_io[16], float *pose_error,
                                int *ok) {
  if (!pe || n_pts < 1 || !xyz || !ref_colors || !new_dIp || !new_cam || !ref_to_new_io)
    return invalid("dsm_pose_estimator_estimate: bad argument");
  dsm_tracker *t = pe->t;
  if (n_pts > t->w * t->h) return invalid("dsm_pose_estimator_estimate: more points than pixels");
  int rc = dsm_tracker_make_k(t, new_cam[0], new_cam[1], new_cam[2], new_cam[3]); // makeK(new_cam), :306
  if (rc) return rc;
  pe->x.resize(n_pts), pe->y.resize(n_pts), pe->z.resize(n_pts);
  for (int i = 0; i < n_pts; i++) { // `float x = pts_[i].first(0)` ..., PoseEstimator.cpp:183-185
    pe->x[i] = (float)xyz[3 * i];
    pe->y[i] = (float)xyz[3 * i + 1];
    pe->z[i] = (float)xyz[3 * i + 2];
  }
  int n[DSM_MAX_LEVELS];
  const float *px[DSM_MAX_LEVELS], *py[DSM_MAX_LEVELS], *pz[DSM_MAX_LEVELS];
  for (int l = 0; l < t->nlevels; l++) {
    n[l] = n_pts; // the same point set on every level, one reference colour per level (:238)
    px[l] = pe->x.data(), py[l] = pe->y.data(), pz[l] = pe->z.data();
  }
  // the float4 template slot holds (x, y, z, refColor[lvl]); ref_aff_g2l_ = (0,0) (:317)
  rc = dsm_tracker_set_ref(t, 0, 0.0, 0.0, ref_ab_exposure, n, px, py, pz, ref_colors);
  if (rc) return rc;
  rc = dsm_tracker_upload_frame(t, DSM_SLOT_NEW_LEFT, new_dIp, new_ab_exposure);
  if (rc) return rc;
  dsm_context *ctx = t->ctx;
  dsm_tracker *ts[1] = {t};
  rc = prepare_batch(ctx, 1, ts, 2);
  if (rc) return rc;
  StartInfo &I = ctx->h_start[0];
  memset(&I, 0, sizeof I);
  se3_from_matrix(ref_to_new_io, I.pose); // SE3(R, t) constructor, :321-322
  for (int l = 0; l < DSM_MAX_LEVELS; l++) I.min_res[l] = std::numeric_limits<double>::quiet_NaN();
  I.scale = 1.0f;
  I.coarsest = coarsest_lvl;
  rc = run_lm_batch(ctx, 1, ts, 2, coarsest_lvl);
  if (rc) return rc;
  const LMState &S = ctx->h_states[0];
  { // refToNew_current.matrix(), :466
    const double x = S.cur[0], y = S.cur[1], z = S.cur[2], w = S.cur[3];
    double *M = ref_to_new_io;
    M[0] = 1 - 2 * (y * y + z * z), M[1] = 2 * (x * y - z * w), M[2] = 2 * (x * z + y * w), M[3] = S.cur[4];
    M[4] = 2 * (x * y + z * w), M[5] = 1 - 2 * (x * x + z * z), M[6] = 2 * (y * z - x * w), M[7] = S.cur[5];
    M[8] = 2 * (x * z - y * w), M[9] = 2 * (y * z + x * w), M[10] = 1 - 2 * (x * x + y * y), M[11] = S.cur[6];
    M[12] = M[13] = M[14] = 0, M[15] = 1;
  }
  const float err = (float)S.last_residuals[0]; // :467
  if (pose_error) *pose_error = err;
  const bool aff_good = S.status == ST_GOOD;                                        // :469-482
  const bool low_res = err < 10.0f;                                                 // RES_THRES, PoseEstimator.h:26
  const int inlier_percent = 100 * float((int)S.last_inners[0]) / (float)n_pts;     // :486
  const bool enough_inlier = inlier_percent > 90;                                   // INNER_PERCENT, PoseEstimator.h:27
  if (ok) *ok = (aff_good && low_res && enough_inlier) ? 1 : 0;                     // :505
  return DSM_OK;
}

// single fused evaluation
static int single_eval(dsm_tracker *t, int mode, int lvl, const double *pose, const double *aff, float scale,
                       float cutoff, SingleOut *out) {
  if (!t) return invalid("null tracker");
  if (lvl < 0 || lvl >= t->nlevels) return invalid("level out of range");
  dsm_context *ctx = t->ctx;
  dsm_tracker *ts[1] = {t};
  int rc = prepare_batch(ctx, 1, ts, mode);
  if (rc) return rc;
  StartInfo &I = ctx->h_start[0];
  memset(&I, 0, sizeof I);
  if (pose) memcpy(I.pose, pose, sizeof I.pose);
  if (aff) memcpy(I.aff, aff, sizeof I.aff);
  I.scale = scale;
  I.cutoff = cutoff;
  I.lvl = lvl;
  DSM_HIP(hipMemcpyAsync(ctx->d_start, ctx->h_start, sizeof(StartInfo), hipMemcpyHostToDevice, ctx->stream));
  launch_lm(ctx->stream, mode, LM_OP_SINGLE_PREP, lvl, 1, ctx->d_tracker_ptrs, ctx->d_states, ctx->d_partials,
            ctx->partial_stride, ctx->d_start, nullptr, nullptr);
  launch_eval(ctx->stream, mode, lvl, round8(level_chunks(t->desc, lvl) > 0 ? level_chunks(t->desc, lvl) : 1), 1, ctx->d_tracker_ptrs,
              ctx->d_states, ctx->d_partials, ctx->partial_stride, nullptr, nullptr);
  launch_lm(ctx->stream, mode, LM_OP_SINGLE_FINISH, lvl, 1, ctx->d_tracker_ptrs, ctx->d_states, ctx->d_partials,
            ctx->partial_stride, nullptr, ctx->d_single, nullptr);
  DSM_HIP(hipGetLastError());
  DSM_HIP(hipMemcpyAsync(ctx->h_single, ctx->d_single, sizeof(SingleOut), hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  *out = ctx->h_single[0];
  return DSM_OK;
}

int dsm_tracker_calc_res_pose(dsm_tracker *t, int lvl, const double pose[7], const double aff[2], float cutoff_th,
                              double rs[6], double H[64], double b[8], int *n_warped) {
  if (!pose || !aff) return invalid("dsm_tracker_calc_res_pose: null pose/aff");
  SingleOut o;
  int rc = single_eval(t, 0, lvl, pose, aff, 1.0f, cutoff_th, &o);
  if (rc) return rc;
  if (rs) memcpy(rs, o.rs, sizeof o.rs);
  if (H) memcpy(H, o.H, sizeof o.H);
  if (b) memcpy(b, o.b, sizeof o.b);
  if (n_warped) *n_warped = o.n_warped;
  return DSM_OK;
}

int dsm_tracker_calc_res_scale(dsm_tracker *t, int lvl, float scale, float cutoff_th, double rs[6], float *H,
                               float *b, int *n_warped) {
  SingleOut o;
  int rc = single_eval(t, 1, lvl, nullptr, nullptr, scale, cutoff_th, &o);
  if (rc) return rc;
  if (rs) memcpy(rs, o.rs, sizeof o.rs);
  if (H) *H = o.Hs;
  if (b) *b = o.bs;
  if (n_warped) *n_warped = o.n_warped;
  return DSM_OK;
}

} // extern "C"
