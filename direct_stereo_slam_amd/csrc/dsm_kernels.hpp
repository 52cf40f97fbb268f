// dsm_kernels.hpp -- host-side launch interface of the HIP kernels (tracker_kernels.hip,
// ringkey_kernels.hip).  Internal to the shared library; the public boundary is
// include/dsm_hotpath.h.
#pragma once
#include "dsm_device.hpp"

namespace dsm {

// Reduction geometry (part of the documented numerics): a workgroup of 256 threads owns a chunk of 256*P
// consecutive template points, thread t handles points chunk*256*P + k*256 + t for k = 0..P-1.  P is chosen per level from its point
// count by one of two tables (dsm_params.chunk_geometry):
//   throughput (0, the default): long chunks -- the per-chunk prologue (three dependent memory round trips before the first point's
//     arithmetic) and epilogue (52 row sums, a barrier, the partial) are paid once per 16 points per thread, and a level of at most 4096
//     points is ONE chunk (the smallest P of 1 / 2 / 4 / 8 / 16 whose 256 P points hold it).  Many problems in flight: the stream, the
//     batched calls.  Round 5's table (16 points per thread from 16 k points up, 8 / 4 / 2 from 4 k / 1 k / 512) measured + 7 % on the
//     streamed bench against the latency table (profiles/r05_ab_geometry.log); round 6's shader-clock stamps inside the tick engine's
//     kernel (profiles/r06_tick_stamps.json) showed its small levels still paying the fixed cost several times over -- a
//     chunk costs about 12 k cycles plus 2 k per point per thread: level 5 of the metric's pyramid (280 points) ran as two workgroups of
//     14.6 k cycles, level 4 (1480) as two of 20 k, level 3 (6688) as four of 28 k; 8.6 % of the kernel's workgroup time for 2.2 % of its bytes
//   latency (1): short chunks -- more workgroups per evaluation, each through sooner: ONE problem in flight (the replay adaptors):
//     0.53 instead of 0.61 ms per frame there
// Results differ between the tables in the last bits of the float sums only (another summation tree); integer outputs are identical.
//   chain (2, round 6): the latency table above 4096 points, ONE chunk up to 4096 (as the throughput table) -- one frame in flight whose
//     small levels run as a chain (an LM loop inside one workgroup: chain_kernel, dsm_params.persistent_coarse < 0; the tick engine's chains)
enum { kGeomThroughput = 0, kGeomLatency = 1, kGeomChain = 2 };
__host__ __device__ inline int pts_per_thread(int n, int geom) {
  if (geom == kGeomLatency || (geom == kGeomChain && n > 4096)) return n >= 256 * 1024 ? 16 : n >= 64 * 1024 ? 8 : n >= 16 * 1024 ? 4 : n >= 4 * 1024 ? 2 : 1;
  return n > 2048 ? 16 : n > 1024 ? 8 : n > 512 ? 4 : n > 256 ? 2 : 1;
}
// chunks of a list of n points at P points per thread
__host__ __device__ inline int chunks_of(int n, int ppt) {
  const int per = kThreads * ppt;
  return (n + per - 1) / per;
}
__host__ __device__ inline int num_chunks(int n, int geom) { return chunks_of(n, pts_per_thread(n, geom)); }

// largest chunk count any n <= cap can produce under either table (P grows with n, so num_chunks is not monotone)
inline int max_chunks_upto(int cap) {
  int best = 0;
  const int edges[8] = {cap, 256 * 1024 - 1, 64 * 1024 - 1, 16 * 1024 - 1, 4 * 1024 - 1, 1023, 511, 255}; // (the latency table's edges; the throughput table's count grows with n)
  for (int g = 0; g < 3; g++)
    for (int e : edges)
      if (e <= cap && num_chunks(e, g) > best) best = num_chunks(e, g);
  return best;
}

enum LMOp {
  LM_OP_STEP = 0,          // consume the evaluation of level `lvl`, advance the LM state machine
  LM_OP_START = 1,         // initialise the state machine at the coarsest level
  LM_OP_SINGLE_PREP = 2,   // dsm_tracker_calc_res_*: build EvalIn from a host supplied pose/scale
  LM_OP_SINGLE_FINISH = 3  // ... and reduce the partials into a SingleOut
};

struct StartInfo { // host -> device, one per problem
  double pose[7];
  double aff[2];
  double min_res[DSM_MAX_LEVELS];
  float scale;
  float cutoff; // single evaluation only
  int coarsest;
  int lvl;      // single evaluation only
};

// mode: 0 = pose (calcResPose + calcGSSSEPose), 1 = scale (calcResScale + calcGSSSEScale)
// rowmap != nullptr: compact launch over nprob rows, row r = problem rowmap[r] (indices relative to trackers / states)
void launch_eval(hipStream_t s, int mode, int lvl, int grid_x, int nprob,
                 const TrackerDev *const *trackers, const LMState *states, float *partials,
                 int partial_stride, int *tickets, int *status_out, bool spec = false, bool split_ro = false,
                 const int *rowmap = nullptr);
void launch_lm(hipStream_t s, int mode, int op, int lvl, int nprob, const TrackerDev *const *trackers,
               LMState *states, const float *partials, int partial_stride, const StartInfo *start,
               SingleOut *single_out, int *status_out, bool spec = false, const int *rowmap = nullptr);

// work-queue kernel (queue_kernel): header of the device-side queue, zeroed before every launch; counters on their
// own 128-byte lines.  Items: (index + 1) << 32 | problem << kQueueChunkBits | chunk.
struct WorkQueue {
  unsigned head, pad0[31];
  unsigned tail, pad1[31];
  unsigned done, pad2[31]; // problems that have terminated
  int error, pad3[31];     // a bounded wait expired
};
constexpr int kQueueChunkBits = 12;
int queue_kernel_blocks_per_cu(int mode);
void launch_queue(hipStream_t s, int mode, int nblocks, int nprob, const TrackerDev *const *trackers, LMState *states,
                  float *partials, int partial_stride, int *tickets, WorkQueue *q, unsigned long long *items, unsigned qmask);

// ---- tick engine of the streaming form (dsm_stream_*, stream_capi.hip) --------------------------------------------
// One TICK advances every resident problem by one LM round, whatever level it stands on: ONE evaluation launch over a
// device-built list of (problem, chunk) items of all levels mixed, then ONE LM launch (a workgroup per slot) that steps the
// problems, builds the next tick's list, retires finished problems into a result array and refills their slots from the
// waiting list -- no host round trip inside an advance of many ticks.
constexpr unsigned kTickNoop = 0x80000000u, kTickCand = 0x40000000u; // item = flags | problem << 12 | chunk
constexpr int kTickChunkBits = 12;
struct TickSegCtl { // one per segment (stream group / companion): item counts of the two lists (tick parity)
  int count[2];
  int overflow; // an item list ran over (cannot happen by construction: checked by the host)
  int chain[2]; // chain entries of the two lists (problems whose pending evaluation is ONE chunk: see tick_eval_kernel), kept behind
                // the list's `cap` item slots
  int pad[27];
};
struct TickModeCtl { // one per problem kind, shared by the kind's segments; every counter is monotonic over the stream's life (64 bits:
                     // 2^31 problems are nine hours at the benchmarked rate)
  long long pending_head, pending_count; // waiting problems: admitted so far (device) / handed over so far (host: the only word it writes)
  long long retired;                     // results written so far
  int ring, pad;                         // entries of the waiting and result rings (a power of two)
  long long sched_evals[DSM_MAX_LEVELS], sched_ro[DSM_MAX_LEVELS]; // evaluations staged (they run in the next tick)
  long long sched_items[DSM_MAX_LEVELS];
};
struct TickPending { // a waiting problem (host -> device)
  StartInfo start;
  const TrackerDev *trk;
  unsigned long long ticket;
};
struct TickResult { // a retired problem (device -> host)
  unsigned long long ticket;
  int status, ticks; // ticks the problem lived (the stream sizes its advances by their mean)
  double cur[7], aff_cur[2];
  double last_residuals[DSM_MAX_LEVELS];
  double flow[3];
  float scale_cur, pad1;
  long long evals[DSM_MAX_LEVELS], evals_ro[DSM_MAX_LEVELS], rounds[DSM_MAX_LEVELS];
};
// start of an advance, before the stream groups fork: which free slot takes which entry of the waiting ring (tick_reserve_kernel)
constexpr int kTickMaxSegs = 20;
struct TickSegDesc {
  int mode, i0, ns;
};
struct TickReserveArgs {
  int nseg;
  TickSegDesc seg[kTickMaxSegs];
};
void launch_tick_reserve(hipStream_t s, const TickReserveArgs &a, const LMState *states, TickModeCtl *mcs, long long *admit_idx);
// the list a tick's LM work appends to (the next tick's evaluations): item slots [0, cap), chain entries [cap, cap + chain_cap)
struct TickList {
  unsigned *items;
  TickSegCtl *seg;
  int buf, cap;
  int chain_cap; // 0: chains off -- every pending evaluation becomes items
  int chain_max_n0; // > 0: only problems whose level-0 template has at most this many points chain (0: all)
};
// what a chain (an evaluation workgroup that also steps its problem, see tick_eval_kernel) needs beyond the evaluation's own arguments
struct TickChainArgs {
  const TrackerDev **trackers;
  LMState *states;
  TickList next;
  TickModeCtl *mc;
  const TickPending *pending;
  TickResult *results;
  unsigned long long *slot_ticket;
  int max_rounds; // LM rounds a chain may run inside one tick
  int flags;      // bit 0: helper wave in the chain's LM step
};
void launch_tick_admit(hipStream_t s, int mode, int nslots, const TrackerDev **trackers, LMState *states, const TickList &list, TickModeCtl *mc,
                       const TickPending *pending, unsigned long long *slot_ticket, const long long *admit_idx);
void launch_tick_eval(hipStream_t s, int mode, int grid, const LMState *states, float *partials, int partial_stride, const unsigned *items,
                      TickSegCtl *seg, int buf, int cap, const TickChainArgs &chain,
                      bool with_chains /* the list may hold chain entries (now or from an earlier setting): the launch must look at them, if only
                                          to turn them into items (chain.max_rounds 0) */);
void launch_tick_lm(hipStream_t s, int mode, int nslots, const TrackerDev **trackers, LMState *states, const float *partials, int partial_stride,
                    const TickList &next, TickModeCtl *mc, const TickPending *pending, TickResult *results, unsigned long long *slot_ticket,
                    int speculate, int opts /* bit 0: helper waves in the LM step, bit 1: the next items' place in the list asked for early */);
// bounded waits of the LM steps' wave hand-shakes that expired since the library was loaded (0 unless something is badly wrong)
int lm_spin_expired();

// persistent LM loop of the small levels on LDS-resident data (levels whose target plane has at most max_px pixels, capped
// by the kernel's LDS arena): coarse_level_fits tells whether a level of w x h pixels and n template points qualifies
void launch_coarse(hipStream_t s, int mode, int nprob, const TrackerDev *const *trackers, LMState *states,
                   int *status_out, int max_px, bool spec);
bool coarse_level_fits(int w, int h, int n, int geom, int max_px);
// the same for the levels whose evaluation is ONE chunk, without LDS staging (chain_kernel: the tick engine's chain as a launch of its
// own, one workgroup per problem, until the problem reaches a level of several chunks or terminates; no speculative candidates)
void launch_chain(hipStream_t s, int mode, int nprob, const TrackerDev *const *trackers, LMState *states, int *status_out);

// row A4 / N3: makeCoarseDepthL0 on the device (template_kernels.hip), batched over the keyframes of a call
struct TplJob {
  int npts, pad;
  const float *pt; // [pu | pv | pidepth | pweight], npts each
  float *ws;       // make_coarse_depth_workspace_floats() floats
  const float *ref[DSM_MAX_LEVELS];
  float4 *pts[DSM_MAX_LEVELS];
  int *d_n; // nlevels + 1 ints: template points per level, then the out-of-bounds flag
};
size_t make_coarse_depth_workspace_floats(int w, int h, int nlevels, int npts);
void launch_make_coarse_depth(hipStream_t s, int w, int h, int nlevels, const TplJob *jobs, int njobs, int max_npts, int texel_floats);

void launch_interleave_template(hipStream_t s, int n, const float *u, const float *v, const float *id,
                                const float *c, float4 *out);
void launch_deinterleave_template(hipStream_t s, int n, const float4 *in, float *u, float *v, float *id,
                                  float *c);
void launch_scale_depth(hipStream_t s, int n, float4 *pts, float scale);
struct ScaleDepthArgs { // every level of one template in one launch (blockIdx.y = level)
  int nlevels;
  int n[DSM_MAX_LEVELS];
  float4 *pts[DSM_MAX_LEVELS];
};
void launch_scale_depth_levels(hipStream_t s, const ScaleDepthArgs &a, int max_n, float scale);
// Template lists are allocated with this many entries of slack (zeroed).  The evaluation loop prefetches template entries up to three
// trips (3 x 256 points) past the end of a chunk; its buffer loads are range-checked against the list (an entry beyond it reads as
// zeros, never used: masked), so the slack is only a second line of defence for the paths that read the list with plain loads
constexpr int kTemplatePad = 1024;
// chunks of an evaluation of level L of a tracker
inline int level_chunks(const TrackerDev &d, int L) { return num_chunks(d.lv[L].n, d.p.geometry); }
// makeImages (upstream DSO): the intensity plane of level 0 from the float image, of level l from level l-1
void launch_pyramid(hipStream_t s, int w, int h, int nlevels, const float *raw, float *const *img);
// the reference's (I, dx, dy) texels out of / into an intensity plane; with d_bad != nullptr import counts the texels whose
// gradient channels are not makeImages' central differences of channel 0 (bitwise, or beyond tol) into d_bad[0] and keeps
// the first such index in d_bad[1]
void launch_dip_export(hipStream_t s, int w, int h, const float *plane, float *out3);
void launch_dip_import(hipStream_t s, int w, int h, const float *in3, float *plane, int *d_bad, float tol);
// one image of a batched hand-over (dsm_upload_images): staged level-0 pixels (float or u8) and the pyramid levels
struct PyrJob {
  const void *raw;
  float *img[DSM_MAX_LEVELS];
  const void *src; // device-visible address of the caller's pinned image (kernel copy path), else null
};
void launch_desc_scatter(hipStream_t s, int n, const TrackerDev *d_src, TrackerDev *const *d_dst);
// raw <- src for every job, rows of row_bytes at pitch `pitch` in src (tight in raw); unit: 16, 4 or 1 bytes per access
void launch_host_rows_copy(hipStream_t s, const PyrJob *d_jobs, int njobs, int row_bytes, int rows, size_t pitch, int unit, int max_blocks);
void launch_pyramid_batched(hipStream_t s, int w, int h, int nlevels, const PyrJob *d_jobs, int njobs, bool u8);

// queue probe of ensure_streams (diag_kernels.hip): a kernel resident for `ticks` of the wall clock, and an empty one
void launch_queue_probe_wait(hipStream_t s, long long ticks);
void launch_queue_probe_empty(hipStream_t s);

// ring-key kernels
void launch_ringkey_knn(hipStream_t s, const float *keysT, int64_t cap, int64_t n_local, int dim,
                        int k, float thres, int shard_rank, int shard_count, const float *d_queries,
                        int nq, unsigned long long *d_scratch, int n_slices,
                        unsigned long long *d_packed_out);
int ringkey_num_slices(int64_t n_local, int nq, int dim);
void launch_ringkey_insert(hipStream_t s, float *keysT, int64_t cap, int64_t pos, int dim,
                           const float *d_key, int nkeys);

} // namespace dsm
