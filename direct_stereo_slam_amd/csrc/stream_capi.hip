// stream_capi.hip -- the streaming (continuous-admission) form of the batched track / scale calls (dsm_stream_*).
//
// dsm_track_and_scale_batch is a synchronous call: every problem of the batch is taken from its first evaluation to its
// last before the call returns, so the lock-step launch schedule runs as many (evaluate, step) rounds per level as the
// SLOWEST problem needs -- 52 rounds on levels 2 and 3 of 512 distinct frames where the median problem needs 6 and 10 --
// and a third of the call is the dependency chain of a few dozen stragglers with the chip nearly idle.  The
// LM loop of a frame is sequential (TrackerAndScaler.cpp:505-593); frames of different sequences are independent
// (FrontEnd.cpp:585-686): the only parallelism there is lies ACROSS frames, and nothing says they must start and end together.
//
// A dsm_stream is a pool of resident LM problems (slots) advanced in PASSES.  One pass = one sweep down the pyramid,
// coarsest level first, with a fixed number of rounds per level -- the rounds MOST problems need (a quantile of what
// recently retired problems took).  A problem that has not finished its level when the pass moves on simply stays where it
// is -- its LMState is resident in device memory and resumable by construction -- and is CARRIED: in the next pass it
// rides with the newly admitted problems' launches at that level, so a straggler's extra rounds cost no launches of their
// own.  Problems retire individually at the end of the pass in which they finish; the slots they free are refilled from the
// waiting queue before the next pass.  No launch in a pass depends on a host read-back (the launch list is fixed when the
// pass starts), the state is read back once per pass.
//
// Same evaluations, same LM steps, same partial order per problem as the batch form: results are bit-identical
// (tests/test_stream.py).  The work-queue kernel -- every problem at its own pace inside one launch -- was the other
// candidate for this; measured with the tail amortised over 2048 frames it runs at 42.7 k frames/s against 48.5 k for the
// launch form (profiles/r04_queue_form_b2048.json, r04_launch_form_b2048.json): its per-item cost is there in steady
// state too, so the streaming form is built on the launches.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <unordered_map>
#include <vector>

#include "dsm_internal.hpp"

using namespace dsm;

namespace {

enum { SLOT_FREE = 0, SLOT_NEW = 1, SLOT_RUNNING = 2 };

struct Slot {
  int state = SLOT_FREE;
  uint64_t ticket = 0;
  dsm_tracker *trk = nullptr;
  int lvl = 0;    // NEW: coarsest level; RUNNING: the level the problem stood at after the last pass
  int passes = 0; // passes it has been resident for
  double pose0[7] = {0, 0, 0, 1, 0, 0, 0}, aff0[2] = {0, 0};
  long long seen_evals[DSM_MAX_LEVELS] = {}, seen_ro[DSM_MAX_LEVELS] = {}; // evaluations already counted in the statistics
};

struct Waiting {
  uint64_t ticket;
  dsm_tracker *trk;
  StartInfo start;
};

struct Seg {
  hipStream_t st;
  int i0, i1, mode;
  bool companion;
  int rows[DSM_MAX_LEVELS];
};

template <typename T>
int alloc_dev(T **p, size_t n) {
  *p = nullptr;
  DSM_HIP(hipMalloc(p, n * sizeof(T)));
  return DSM_OK;
}
template <typename T>
int alloc_pinned(T **p, size_t n) {
  *p = nullptr;
  DSM_HIP(hipHostMalloc(p, n * sizeof(T), hipHostMallocDefault));
  return DSM_OK;
}

int invalid(const char *msg) {
  set_error(msg);
  return DSM_ERR_INVALID;
}

constexpr size_t kHistCap = 4096; // retired problems whose round counts the schedule looks at

} // namespace

struct dsm_stream {
  dsm_context *ctx = nullptr;
  int cap[2] = {0, 0}; // slots of mode 0 (trackNewestCoarse) / mode 1 (optimizeScale); track slots come first
  int N = 0;
  int w = 0, h = 0, nlevels = 0; // geometry of the first tracker submitted; every tracker of the stream must share it
  int partial_stride = 0;
  TrackerDev **d_tracker_ptrs = nullptr, **h_tracker_ptrs = nullptr;
  LMState *d_states = nullptr, *h_states = nullptr;
  float *d_partials = nullptr;
  StartInfo *d_start = nullptr, *h_start = nullptr;
  int *d_status = nullptr;
  int *d_tickets = nullptr;
  int *d_rowmap = nullptr, *h_rowmap = nullptr; // [DSM_MAX_LEVELS + 1][N]: rows per level (relative to their segment), then the new slots (relative to their mode's base)
  std::vector<Slot> slots;
  std::deque<Waiting> waiting[2];
  std::deque<dsm_stream_result> done;
  uint64_t next_ticket = 1;
  double quantile[DSM_MAX_LEVELS] = {0.75, 0.75, 0.75, 0.75, 0.75, 0.75};
  int fixed_rounds[2][DSM_MAX_LEVELS] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}; // > 0: overrides the learnt schedule
  std::vector<int> hist[2][DSM_MAX_LEVELS];
  size_t hist_pos[2] = {0, 0};
  int rounds[2][DSM_MAX_LEVELS] = {{6, 8, 10, 12, 16, 16}, {4, 4, 4, 4, 4, 4}};
  int rounds_max[2][DSM_MAX_LEVELS] = {{12, 22, 52, 52, 52, 52}, {8, 8, 8, 8, 8, 8}}; // the most any retired problem needed (+ 1)
  dsm_stats stats[2]{};
  long long passes = 0, retired[2] = {0, 0}, carried_slot_passes = 0;
  // ---- tick engine (engine 1): every resident problem advances one LM round per tick, admission and retirement on the device ----
  int engine = 1;
  int ticks = 32; // ticks per advance (one host read-back per advance)
  bool tick_ready = false;
  std::vector<Seg> tsegs;
  TickSegCtl *d_segctl = nullptr, *h_segctl = nullptr;
  TickModeCtl *d_modectl = nullptr, *h_modectl = nullptr;
  std::vector<unsigned *> d_items; // [segment][2]
  std::vector<int> items_cap, last_count, last_chain;
  // chains (tick_eval_kernel): LM rounds a problem whose pending evaluation is ONE chunk may run inside one tick, in the workgroup that
  // evaluates it; 0 = off (every round costs a tick).  A scheduling switch: results are bit-identical.
  int chain_rounds = DSM_STREAM_CHAIN_DEFAULT;
  bool chain_possible = false; // a problem the chains apply to was submitted at some time (else the evaluation launches go without the chains' code)
  int chain_max_n0 = DSM_STREAM_CHAIN_DEFAULT_MAX_N0; // the default chains semi-dense problems only; a number named by the caller: every problem
  // options of the tick kernels' LM step, both measured on (profiles/r06b_ab_lm_step.log, where they were switched through the environment):
  int chain_flags = 1; // bit 0: helper wave in a chain's LM step
  int lm_opts = 3;     // tick_lm_kernel: bit 0 helper waves, bit 1 the next items' place in the list reserved before the solve
  TickPending *d_pending[2] = {nullptr, nullptr}, *h_pending[2] = {nullptr, nullptr}; // waiting rings (device / pinned staging)
  TickResult *d_results[2] = {nullptr, nullptr}, *h_results[2] = {nullptr, nullptr};   // result rings (device / two pinned copies)
  int ring[2] = {0, 0};
  long long *h_count_word = nullptr; // pinned: the pending_count words of the two advances in flight
  hipEvent_t ev_begin[2] = {nullptr, nullptr}, ev_end[2] = {nullptr, nullptr};
  bool pipelined = true;       // advance returns once its work is enqueued; results surface one advance late
  // ticks per advance.  auto_ticks (the default until dsm_stream_set_engine names a number): as many as retire 7/8 of what the
  // advance hands over -- handed x (mean life of a problem in ticks, learnt from the retired ones) / slots x 7/8 (tick_advance)
  bool auto_ticks = true;
  double life_ticks = 0.0;
  int inflight_ticks[2] = {0, 0};
  long long collected = 0;     // advances whose read-back has been processed
  long long handed[2] = {0, 0}; // problems appended to the device's waiting ring so far
  struct Seen {
    long long admitted, retired;
    long long evals[DSM_MAX_LEVELS], ro[DSM_MAX_LEVELS];
  } seen[2];
  int inflight_parity[2] = {0, 0};
  bool inflight_timed[2] = {false, false}; // the advance was enqueued with per-dispatch events (dsm_context_set_timing)
  std::vector<int> ev_lvl;
  // scheduling switches of the problems in flight, BY VALUE (a tracker may be destroyed as soon as its own result is back -- ADVICE r04:
  // the stream used to keep a pointer into the first submitted tracker's params for its whole life); refreshed at every hand-over
  bool have_sched = false;
  int sched_fixed_schedule = 0, sched_speculate = 0;
  unsigned long long *d_slot_ticket = nullptr;
  long long *d_admit_idx = nullptr; // per slot: the waiting-ring entry it takes at the start of this advance, or -1 (tick_reserve_kernel)
  int parity = 0;
  int resident[2] = {0, 0};
  struct Origin {
    dsm_tracker *trk;
    double pose0[7], aff0[2];
    long long admitted_at; // advance in which the problem was handed to the device (-1: still waiting)
  };
  std::unordered_map<uint64_t, Origin> origin;
  long long advances = 0, total_ticks = 0;
};

// the tick engine's resources (tick_setup); safe on a partially set-up stream, leaves every pointer null
static void tick_free(dsm_stream *s) {
  auto dev = [](auto *&p) { if (p) hipFree(p); p = nullptr; };
  auto pin = [](auto *&p) { if (p) hipHostFree(p); p = nullptr; };
  dev(s->d_segctl), pin(s->h_segctl), dev(s->d_modectl), pin(s->h_modectl);
  for (unsigned *&p : s->d_items) dev(p);
  s->d_items.clear();
  for (int m = 0; m < 2; m++) dev(s->d_pending[m]), pin(s->h_pending[m]), dev(s->d_results[m]), pin(s->h_results[m]);
  dev(s->d_slot_ticket), dev(s->d_admit_idx), pin(s->h_count_word);
  for (int k = 0; k < 2; k++) {
    if (s->ev_begin[k]) hipEventDestroy(s->ev_begin[k]);
    if (s->ev_end[k]) hipEventDestroy(s->ev_end[k]);
    s->ev_begin[k] = s->ev_end[k] = nullptr;
  }
  s->tick_ready = false;
}

static void stream_free(dsm_stream *s) {
  hipFree(s->d_tracker_ptrs);
  if (s->h_tracker_ptrs) hipHostFree(s->h_tracker_ptrs);
  hipFree(s->d_states);
  if (s->h_states) hipHostFree(s->h_states);
  hipFree(s->d_partials);
  hipFree(s->d_start);
  if (s->h_start) hipHostFree(s->h_start);
  hipFree(s->d_status);
  hipFree(s->d_tickets);
  hipFree(s->d_rowmap);
  if (s->h_rowmap) hipHostFree(s->h_rowmap);
  tick_free(s);
  delete s;
}

// the partials need the trackers' geometry: allocated on the first submission
static int stream_bind_geometry(dsm_stream *s, dsm_tracker *t) {
  if (s->partial_stride) {
    if (t->w != s->w || t->h != s->h || t->nlevels != s->nlevels) return invalid("dsm_stream: all trackers of a stream must share image size and levels");
    return DSM_OK;
  }
  DSM_HIP(hipSetDevice(s->ctx->device));
  s->w = t->w, s->h = t->h, s->nlevels = t->nlevels;
  const int ps = 2 * max_chunks_upto(t->w * t->h) * kPartialStride; // second half: the speculative candidate's partials
  int rc = alloc_dev(&s->d_partials, (size_t)s->N * ps);
  if (rc) return rc;
  s->partial_stride = ps;
  return DSM_OK;
}

static int tick_advance(dsm_stream *s);
static int tick_sync(dsm_stream *s);

extern "C" {

int dsm_stream_create(dsm_context *ctx, int track_slots, int scale_slots, dsm_stream **out) {
  if (!ctx || !out || track_slots < 0 || scale_slots < 0 || track_slots + scale_slots < 1) return invalid("dsm_stream_create: bad argument");
  *out = nullptr;
  DSM_HIP(hipSetDevice(ctx->device));
  dsm_stream *s = new dsm_stream();
  s->ctx = ctx;
  s->cap[0] = track_slots, s->cap[1] = scale_slots;
  const int N = s->N = track_slots + scale_slots;
  s->slots.resize(N);
  int rc = DSM_OK;
  if (!rc) rc = alloc_dev(&s->d_tracker_ptrs, N);
  if (!rc) rc = alloc_pinned(&s->h_tracker_ptrs, N);
  if (!rc) rc = alloc_dev(&s->d_states, N);
  if (!rc) rc = alloc_pinned(&s->h_states, N);
  if (!rc) rc = alloc_dev(&s->d_start, N);
  if (!rc) rc = alloc_pinned(&s->h_start, N);
  if (!rc) rc = alloc_dev(&s->d_status, 2 * (size_t)N);
  if (!rc) rc = alloc_dev(&s->d_tickets, N);
  if (!rc) rc = alloc_dev(&s->d_rowmap, (size_t)(DSM_MAX_LEVELS + 1) * N);
  if (!rc) rc = alloc_pinned(&s->h_rowmap, (size_t)(DSM_MAX_LEVELS + 1) * N);
  if (!rc) {
    hipError_t e = hipMemsetAsync(s->d_tickets, 0, sizeof(int) * N, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_states, 0, sizeof(LMState) * N, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "dsm_stream_create: clearing the slot arrays", __FILE__, __LINE__);
  }
  if (rc) {
    stream_free(s);
    return rc;
  }
  memset(s->h_tracker_ptrs, 0, sizeof(TrackerDev *) * N);
  memset(s->h_start, 0, sizeof(StartInfo) * N);
  *out = s;
  return DSM_OK;
}

int dsm_stream_destroy(dsm_stream *s) {
  if (!s) return DSM_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  for (hipStream_t st : s->ctx->extra_streams) hipStreamSynchronize(st);
  if (s->ctx->companion_stream) hipStreamSynchronize(s->ctx->companion_stream);
  stream_free(s);
  return DSM_OK;
}

int dsm_stream_set_quantile(dsm_stream *s, int lvl, double q) {
  if (!s || !(q > 0.0 && q <= 1.0) || lvl >= DSM_MAX_LEVELS) return invalid("dsm_stream_set_quantile: level < DSM_MAX_LEVELS (negative: all), 0 < q <= 1");
  for (int l = 0; l < DSM_MAX_LEVELS; l++)
    if (lvl < 0 || l == lvl) s->quantile[l] = q;
  return DSM_OK;
}

int dsm_stream_set_engine(dsm_stream *s, int engine, int ticks_per_advance) {
  if (!s || engine < 0 || engine > 1 || ticks_per_advance < -1 || ticks_per_advance > 4096)
    return invalid("dsm_stream_set_engine: engine 0 / 1, ticks -1 (the stream's own choice), 0 (keep), 1 .. 4096");
  if (engine != s->engine) {
    // both engines share the slot states: a switch needs the stream EMPTY on the device -- every advance read back (the pipelined
    // counts are one advance old), nothing resident, nothing handed to the device's waiting ring and not yet admitted
    if (s->engine == 1 && s->tick_ready) {
      DSM_HIP(hipSetDevice(s->ctx->device));
      const int rc = tick_sync(s);
      if (rc) return rc;
    }
    int resident = 0;
    dsm_stream_counts(s, &resident, nullptr, nullptr);
    const bool on_the_way = s->handed[0] != s->seen[0].admitted || s->handed[1] != s->seen[1].admitted || s->collected != s->advances;
    if (resident || on_the_way) return invalid("dsm_stream_set_engine: problems are resident or on their way to the device (drain the stream first)");
  }
  s->engine = engine;
  if (ticks_per_advance > 0) s->ticks = ticks_per_advance, s->auto_ticks = false;
  if (ticks_per_advance < 0) s->auto_ticks = true;
  return DSM_OK;
}

int dsm_stream_set_chain(dsm_stream *s, int max_rounds) {
  if (!s || max_rounds < -1 || max_rounds > 4096) return invalid("dsm_stream_set_chain: rounds 0 (off) .. 4096, or -1 (the default)");
  s->chain_rounds = max_rounds < 0 ? DSM_STREAM_CHAIN_DEFAULT : max_rounds; // (takes effect with the next advance: the lists of both kinds are read every tick)
  s->chain_max_n0 = max_rounds < 0 ? DSM_STREAM_CHAIN_DEFAULT_MAX_N0 : 0;
  if (max_rounds > 0) s->chain_possible = true; // (a bound named by the caller applies to every problem)
  return DSM_OK;
}

int dsm_stream_set_rounds(dsm_stream *s, int mode, const int *rounds_per_level) {
  if (!s || mode < 0 || mode > 1) return invalid("dsm_stream_set_rounds: bad argument");
  for (int l = 0; l < DSM_MAX_LEVELS; l++) s->fixed_rounds[mode][l] = rounds_per_level ? rounds_per_level[l] : 0;
  return DSM_OK;
}

static int submit_common(dsm_stream *s, int mode, int n, dsm_tracker *const *ts, int coarsest, uint64_t *tickets_out) {
  if (!s || n < 0 || (n && !ts)) return invalid("dsm_stream_submit: bad argument");
  if (s->cap[mode] == 0 && n) return invalid("dsm_stream_submit: the stream has no slots of this kind");
  for (int i = 0; i < n; i++) {
    dsm_tracker *t = ts[i];
    if (!t || t->ctx != s->ctx) return invalid("dsm_stream_submit: tracker does not belong to the stream's context");
    int rc = stream_bind_geometry(s, t);
    if (rc) return rc;
    if (coarsest < 0 || coarsest >= t->nlevels) return invalid("coarsest level out of range"); // :457 / :856
    rc = check_ready(t, mode);
    if (rc) return rc;
    if (s->chain_max_n0 <= 0 || t->desc.lv[0].n <= s->chain_max_n0) s->chain_possible = true;
  }
  (void)tickets_out;
  return DSM_OK;
}

int dsm_stream_submit_track(dsm_stream *s, int n, dsm_tracker *const *ts, const double *pose0, const double *aff0, int coarsest_lvl,
                            const double *min_res_for_abort, uint64_t *tickets_out) {
  if (n && (!pose0 || !aff0)) return invalid("dsm_stream_submit_track: null pose/aff");
  int rc = submit_common(s, 0, n, ts, coarsest_lvl, tickets_out);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    Waiting wq;
    wq.ticket = s->next_ticket++;
    wq.trk = ts[i];
    StartInfo &I = wq.start;
    memset(&I, 0, sizeof I);
    memcpy(I.pose, pose0 + 7 * i, sizeof I.pose);
    memcpy(I.aff, aff0 + 2 * i, sizeof I.aff);
    for (int l = 0; l < DSM_MAX_LEVELS; l++)
      I.min_res[l] = min_res_for_abort ? min_res_for_abort[DSM_MAX_LEVELS * i + l] : std::numeric_limits<double>::quiet_NaN();
    I.scale = 1.0f;
    I.coarsest = coarsest_lvl;
    if (tickets_out) tickets_out[i] = wq.ticket;
    dsm_stream::Origin og;
    og.trk = ts[i], og.admitted_at = -1;
    memcpy(og.pose0, I.pose, sizeof og.pose0);
    memcpy(og.aff0, I.aff, sizeof og.aff0);
    s->origin[wq.ticket] = og;
    s->waiting[0].push_back(wq);
  }
  return DSM_OK;
}

int dsm_stream_submit_scale(dsm_stream *s, int n, dsm_tracker *const *ts, const float *scale0, int coarsest_lvl, uint64_t *tickets_out) {
  if (n && !scale0) return invalid("dsm_stream_submit_scale: null scale");
  int rc = submit_common(s, 1, n, ts, coarsest_lvl, tickets_out);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    Waiting wq;
    wq.ticket = s->next_ticket++;
    wq.trk = ts[i];
    StartInfo &I = wq.start;
    memset(&I, 0, sizeof I);
    I.pose[3] = 1.0;
    for (int l = 0; l < DSM_MAX_LEVELS; l++) I.min_res[l] = std::numeric_limits<double>::quiet_NaN();
    I.scale = scale0[i];
    I.coarsest = coarsest_lvl;
    if (tickets_out) tickets_out[i] = wq.ticket;
    dsm_stream::Origin og;
    memset(&og, 0, sizeof og);
    og.trk = ts[i], og.admitted_at = -1;
    s->origin[wq.ticket] = og;
    s->waiting[1].push_back(wq);
  }
  return DSM_OK;
}

int dsm_stream_counts(dsm_stream *s, int *resident_out, int *waiting_out, int *results_out) {
  if (!s) return invalid("null stream");
  int r = 0;
  for (const Slot &sl : s->slots) r += sl.state != SLOT_FREE;
  long long on_the_way = 0;
  if (s->engine == 1) { // as of the last read-back: in a slot / in the device's waiting ring
    r = (int)((s->seen[0].admitted - s->seen[0].retired) + (s->seen[1].admitted - s->seen[1].retired));
    on_the_way = (s->handed[0] - s->seen[0].admitted) + (s->handed[1] - s->seen[1].admitted);
  }
  if (resident_out) *resident_out = r;
  if (waiting_out) *waiting_out = (int)(s->waiting[0].size() + s->waiting[1].size() + on_the_way);
  if (results_out) *results_out = (int)s->done.size();
  return DSM_OK;
}

int dsm_stream_results(dsm_stream *s, int max_results, dsm_stream_result *out, int *n_out) {
  if (!s || max_results < 0 || (max_results && !out) || !n_out) return invalid("dsm_stream_results: bad argument");
  int k = 0;
  while (k < max_results && !s->done.empty()) {
    out[k++] = s->done.front();
    s->done.pop_front();
  }
  *n_out = k;
  return DSM_OK;
}

int dsm_stream_get_stats(dsm_stream *s, dsm_stats *track_out, dsm_stats *scale_out) {
  if (!s) return invalid("null stream");
  if (track_out) *track_out = s->stats[0];
  if (scale_out) *scale_out = s->stats[1];
  return DSM_OK;
}

int dsm_stream_get_schedule(dsm_stream *s, int mode, int *rounds_out, long long *passes_out, long long *retired_out, long long *carried_out) {
  if (!s || mode < 0 || mode > 1) return invalid("dsm_stream_get_schedule: bad argument");
  if (rounds_out)
    for (int l = 0; l < DSM_MAX_LEVELS; l++) rounds_out[l] = s->rounds[mode][l];
  if (passes_out) *passes_out = s->passes;
  if (retired_out) *retired_out = s->retired[mode];
  if (carried_out) *carried_out = s->carried_slot_passes;
  return DSM_OK;
}

// One pass.  (i) waiting problems are admitted into free slots, (ii) their state machines are started, (iii) every level
// from the coarsest a resident problem stands on down to level 0 gets its rounds -- compact launches over the slots that
// can be at that level in this pass: the new ones and the carried ones standing at or above it -- (iv) the states are read
// back once; problems that terminated retire (dsm_stream_results), the others are carried.
int dsm_stream_advance(dsm_stream *s) {
  if (!s) return invalid("null stream");
  if (s->engine == 1) return tick_advance(s);
  dsm_context *ctx = s->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  const int N = s->N, cap0 = s->cap[0];
  int ng = ctx->n_streams < 1 ? 1 : ctx->n_streams;
  if (ng > cap0) ng = cap0 > 0 ? cap0 : 1;
  // ---- (i) admission: a free slot of the problem's kind, in the stream group with the fewest resident problems ----
  std::vector<Seg> segs;
  if (cap0 > 0)
    for (int g = 0; g < ng; g++) {
      const int g0 = (int)((long long)cap0 * g / ng), g1 = (int)((long long)cap0 * (g + 1) / ng);
      if (g1 > g0) segs.push_back(Seg{nullptr, g0, g1, 0, false, {}});
    }
  const int n_track_segs = (int)segs.size();
  if (s->cap[1] > 0) segs.push_back(Seg{nullptr, cap0, N, 1, true, {}});
  std::vector<dsm_tracker *> fresh;
  int n_new[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++) {
    std::deque<Waiting> &wq = s->waiting[mode];
    if (wq.empty()) continue;
    const int sg0 = mode == 0 ? 0 : n_track_segs, sg1 = mode == 0 ? n_track_segs : (int)segs.size();
    std::vector<int> live(segs.size(), 0), cursor(segs.size(), 0);
    for (int si = sg0; si < sg1; si++) {
      cursor[si] = segs[si].i0;
      for (int i = segs[si].i0; i < segs[si].i1; i++) live[si] += s->slots[i].state != SLOT_FREE;
    }
    while (!wq.empty()) {
      int best = -1;
      for (int si = sg0; si < sg1; si++)
        if (live[si] < segs[si].i1 - segs[si].i0 && (best < 0 || live[si] < live[best])) best = si;
      if (best < 0) break; // every slot of this kind is taken
      int &c = cursor[best];
      while (s->slots[c].state != SLOT_FREE) c++;
      const Waiting &wt = wq.front();
      Slot &sl = s->slots[c];
      sl = Slot();
      sl.state = SLOT_NEW;
      sl.ticket = wt.ticket;
      sl.trk = wt.trk;
      sl.lvl = wt.start.coarsest;
      memcpy(sl.pose0, wt.start.pose, sizeof sl.pose0);
      memcpy(sl.aff0, wt.start.aff, sizeof sl.aff0);
      s->h_start[c] = wt.start;
      s->h_tracker_ptrs[c] = wt.trk->d_desc;
      s->h_rowmap[(size_t)DSM_MAX_LEVELS * N + (mode == 0 ? 0 : cap0) + n_new[mode]] = c - (mode == 0 ? 0 : cap0);
      n_new[mode]++;
      fresh.push_back(wt.trk);
      live[best]++;
      wq.pop_front();
    }
  }
  int n_live = 0, top = -1, top2 = -1;
  const dsm_params *P = nullptr;
  for (int i = 0; i < N; i++) {
    const Slot &sl = s->slots[i];
    if (sl.state == SLOT_FREE) continue;
    n_live++;
    if (!P) P = &sl.trk->params;
    if (i < cap0)
      top = sl.lvl > top ? sl.lvl : top;
    else
      top2 = sl.lvl > top2 ? sl.lvl : top2;
  }
  if (n_live == 0) return DSM_OK;
  // Nothing is waiting and the pool is at most a quarter full: what is resident is the END of the job -- nothing rides with
  // the stragglers any more, so a pass gives every level the rounds the slowest retired problem needed (they finish in one or
  // two passes instead of one level per pass) and evaluates and steps in one fused launch per round.
  const bool draining = s->waiting[0].empty() && s->waiting[1].empty() && 4 * n_live <= N;
  const int nlevels = s->nlevels;
  int rc = ensure_streams(ctx, ng, s->cap[1] > 0);
  if (rc) return rc;
  for (int si = 0; si < (int)segs.size(); si++) // (a stream without track slots runs its scale segment on the context's stream)
    segs[si].st = si == 0 ? ctx->stream : segs[si].companion ? ctx->companion_stream : ctx->extra_streams[si - 1];
  if (!fresh.empty()) {
    rc = sync_descs(ctx, fresh.data(), (int)fresh.size());
    if (rc) return rc;
  }
  DSM_HIP(hipEventRecord(ctx->ev_total[0], ctx->stream));
  // ---- row maps: level L of segment sg = its slots that can stand on level L during this pass ----
  int grid_x[DSM_MAX_LEVELS], level_pts[DSM_MAX_LEVELS];
  for (int L = 0; L < nlevels; L++) {
    int max_chunks = 1, max_n = 0;
    for (int i = 0; i < N; i++) {
      const Slot &sl = s->slots[i];
      if (sl.state == SLOT_FREE) continue;
      const int n_l = sl.trk->desc.lv[L].n, c = level_chunks(sl.trk->desc, L);
      if (c > max_chunks) max_chunks = c;
      if (n_l > max_n) max_n = n_l;
    }
    grid_x[L] = max_chunks < 8 ? max_chunks : (max_chunks + 7) & ~7;
    level_pts[L] = max_n;
    for (Seg &sg : segs) {
      int r = 0;
      for (int i = sg.i0; i < sg.i1; i++)
        if (s->slots[i].state != SLOT_FREE && s->slots[i].lvl >= L) s->h_rowmap[(size_t)L * N + sg.i0 + r++] = i - sg.i0;
      sg.rows[L] = r;
    }
  }
  DSM_HIP(hipMemcpyAsync(s->d_rowmap, s->h_rowmap, sizeof(int) * (size_t)(DSM_MAX_LEVELS + 1) * N, hipMemcpyHostToDevice, ctx->stream));
  // ---- (ii) start the admitted problems (LM_OP_START over the list of new slots of each kind) ----
  if (n_new[0] + n_new[1] > 0) {
    DSM_HIP(hipMemcpyAsync(s->d_tracker_ptrs, s->h_tracker_ptrs, sizeof(TrackerDev *) * N, hipMemcpyHostToDevice, ctx->stream));
    DSM_HIP(hipMemcpyAsync(s->d_start, s->h_start, sizeof(StartInfo) * N, hipMemcpyHostToDevice, ctx->stream));
    for (int mode = 0; mode < 2; mode++) {
      if (!n_new[mode]) continue;
      const int base = mode == 0 ? 0 : cap0;
      launch_lm(ctx->stream, mode, LM_OP_START, 0, n_new[mode], s->d_tracker_ptrs + base, s->d_states + base,
                s->d_partials + (size_t)base * s->partial_stride, s->partial_stride, s->d_start + base, nullptr, s->d_status + 2 * base, false,
                s->d_rowmap + (size_t)DSM_MAX_LEVELS * N + base);
    }
  }
  // ---- (iii) the pass ----
  size_t ev_used = 0;
  std::vector<int> ev_lvl;
  auto launch_round = [&](const Seg &sg, int L, int k) -> int {
    const int rows = sg.rows[L];
    if (rows == 0) return DSM_OK;
    const int *rowmap = s->d_rowmap + (size_t)L * N + sg.i0;
    // the speculative second candidate, the fused LM step and the split of the residual-only evaluations: the rules of
    // run_lm_batch (dsm_capi.hip), applied to this launch's rows
    const bool spec = P->fixed_schedule <= 0 && (P->speculate >= 2 || (P->speculate == 1 && level_pts[L] <= 8192 && (long long)rows * level_pts[L] <= 1000000ll));
    const bool fused = L > 0 && (P->fuse_lm >= 2 || (P->fuse_lm == 1 && (rows <= 8 || (draining && rows <= 128))));
    const bool split_ro = !sg.companion && !fused && k > 0 && level_pts[L] >= 100000 && (long long)rows * level_pts[L] >= 8000000ll;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (ctx->timing && !sg.companion) {
      ea = get_event(ctx, ev_used++);
      eb = get_event(ctx, ev_used++);
      ev_lvl.push_back(L);
      if (ea) DSM_HIP(hipEventRecord(ea, sg.st));
    }
    float *part = s->d_partials + (size_t)sg.i0 * s->partial_stride;
    launch_eval(sg.st, sg.mode, L, grid_x[L], rows, s->d_tracker_ptrs + sg.i0, s->d_states + sg.i0, part, s->partial_stride,
                fused ? s->d_tickets + sg.i0 : nullptr, s->d_status + 2 * sg.i0, spec, split_ro, rowmap);
    if (eb) DSM_HIP(hipEventRecord(eb, sg.st));
    if (!fused)
      launch_lm(sg.st, sg.mode, LM_OP_STEP, L, rows, s->d_tracker_ptrs + sg.i0, s->d_states + sg.i0, part, s->partial_stride, nullptr, nullptr,
                s->d_status + 2 * sg.i0, spec, rowmap);
    return DSM_OK;
  };
  if (segs.size() > 1) { // fork: the other segments' streams start behind the uploads and the START launches
    DSM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
    for (size_t si = 1; si < segs.size(); si++) DSM_HIP(hipStreamWaitEvent(segs[si].st, ctx->fork_event, 0));
  }
  for (int L = top > top2 ? top : top2; L >= 0; L--) {
    auto seg_rounds = [&](const Seg &sg) {
      int r = s->fixed_rounds[sg.mode][L] > 0 ? s->fixed_rounds[sg.mode][L] : draining ? s->rounds_max[sg.mode][L] : s->rounds[sg.mode][L];
      if (P->fixed_schedule > 0) r = 1 + P->fixed_schedule; // the benchmark schedule: every problem needs exactly 1 + K rounds
      return r < 1 ? 1 : r;
    };
    int kmax = 0;
    for (const Seg &sg : segs)
      if (sg.rows[L] > 0 && seg_rounds(sg) > kmax) kmax = seg_rounds(sg);
    for (int k = 0; k < kmax; k++)
      for (int si = (int)segs.size() - 1; si >= 0; si--) { // (the companion's round first, as in run_lm_batch)
        const Seg &sg = segs[si];
        if (k < seg_rounds(sg)) {
          rc = launch_round(sg, L, k);
          if (rc) return rc;
        }
      }
    for (const Seg &sg : segs)
      if (sg.rows[L] > 0) s->stats[sg.mode].launches[L] += seg_rounds(sg);
  }
  DSM_HIP(hipGetLastError());
  for (size_t si = 1; si < segs.size(); si++) { // join
    hipEvent_t ev = segs[si].companion ? ctx->companion_event : ctx->join_events[si - 1];
    DSM_HIP(hipEventRecord(ev, segs[si].st));
    DSM_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
  }
  // ---- (iv) one read-back ----
  DSM_HIP(hipMemcpyAsync(s->h_states, s->d_states, sizeof(LMState) * N, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipEventRecord(ctx->ev_total[1], ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  s->passes++;
  float ms = 0;
  DSM_HIP(hipEventElapsedTime(&ms, ctx->ev_total[0], ctx->ev_total[1]));
  s->stats[0].total_ms += ms, s->stats[1].total_ms += ms;
  s->stats[0].polls += 1;
  if (ctx->timing) collect_eval_timing(ctx, ev_lvl, nlevels, s->stats[0]);
  for (int i = 0; i < N; i++) {
    Slot &sl = s->slots[i];
    if (sl.state == SLOT_FREE) continue;
    const int mode = i < cap0 ? 0 : 1;
    const LMState &S = s->h_states[i];
    dsm_stats &st = s->stats[mode];
    sl.passes++;
    for (int l = 0; l < nlevels; l++) { // what THIS pass evaluated
      const long long de = S.evals[l] - sl.seen_evals[l], dr = S.evals_ro[l] - sl.seen_ro[l];
      sl.seen_evals[l] = S.evals[l], sl.seen_ro[l] = S.evals_ro[l];
      st.evals[l] += de;
      st.evals_residual_only[l] += dr;
      const long long nl = sl.trk->desc.lv[l].n, img = 12ll * (sl.trk->w >> l) * (sl.trk->h >> l);
      st.algorithmic_bytes += de * (16ll * nl + (48ll * nl < img ? 48ll * nl : img)); // SURVEY.md 8d, as run_lm_batch
    }
    if (S.status == ST_RUNNING) {
      if (sl.passes > 4096) {
        set_error("internal: a resident problem did not terminate within 4096 passes");
        return DSM_ERR_STATE;
      }
      sl.state = SLOT_RUNNING;
      sl.lvl = S.lvl;
      s->carried_slot_passes++;
      continue;
    }
    dsm_stream_result r;
    memset(&r, 0, sizeof r);
    r.ticket = sl.ticket;
    r.kind = mode;
    r.status = S.status;
    r.passes = sl.passes;
    if (mode == 0) {
      // the reference writes lastToNew_out / aff_g2l_out at :612-613, i.e. also when the later affine plausibility checks
      // (:615-626) fail, but not when a level aborts (:598): as dsm_track_batch
      const bool wrote = S.status == ST_GOOD || S.status == ST_BAD_AFFINE;
      memcpy(r.pose, wrote ? S.cur : sl.pose0, sizeof r.pose);
      memcpy(r.aff, wrote ? S.aff_cur : sl.aff0, sizeof r.aff);
      r.good = S.status == ST_GOOD ? 1 : 0;
      memcpy(r.flow, S.flow, sizeof r.flow);
      r.scale = 1.0f;
    } else {
      r.good = 1;
      r.scale = S.scale_cur;                    // :954
      r.err = (float)S.last_residuals[0];       // :963
      r.pose[3] = 1.0;
    }
    memcpy(r.last_residuals, S.last_residuals, sizeof r.last_residuals);
    for (int l = 0; l < DSM_MAX_LEVELS; l++) r.evals[l] = S.evals[l];
    s->done.push_back(r);
    s->origin.erase(sl.ticket);
    s->retired[mode]++;
    for (int l = 0; l < nlevels; l++) {
      std::vector<int> &hv = s->hist[mode][l];
      if (hv.size() < kHistCap)
        hv.push_back((int)S.rounds[l]);
      else
        hv[s->hist_pos[mode] % kHistCap] = (int)S.rounds[l];
    }
    s->hist_pos[mode]++;
    sl = Slot();
  }
  // ---- the next pass's rounds per level: the quantile of what the recently retired problems needed ----
  for (int mode = 0; mode < 2; mode++) {
    if (s->hist[mode][0].size() < 16) continue;
    std::vector<int> r;
    for (int l = 0; l < nlevels; l++) {
      r = s->hist[mode][l];
      size_t k = (size_t)(s->quantile[l] * (double)(r.size() - 1) + 0.5);
      if (k >= r.size()) k = r.size() - 1;
      std::nth_element(r.begin(), r.begin() + k, r.end());
      s->rounds[mode][l] = r[k] < 1 ? 1 : r[k];
      s->rounds_max[mode][l] = *std::max_element(r.begin(), r.end()) + 1;
    }
  }
  return DSM_OK;
}


} // extern "C"

// ---- tick engine -------------------------------------------------------------------------------------------------------
template <typename T>
static int regrow_dev(T **p, size_t n) {
  if (*p) DSM_HIP(hipFree(*p));
  return alloc_dev(p, n);
}
template <typename T>
static int regrow_pinned(T **p, size_t n) {
  if (*p) DSM_HIP(hipHostFree(*p));
  return alloc_pinned(p, n);
}

static int tick_setup(dsm_stream *s) {
  dsm_context *ctx = s->ctx;
  const int cap0 = s->cap[0], N = s->N;
  int ng = ctx->n_streams < 1 ? 1 : ctx->n_streams;
  if (ng > cap0) ng = cap0 > 0 ? cap0 : 1;
  s->tsegs.clear();
  if (cap0 > 0)
    for (int g = 0; g < ng; g++) {
      const int g0 = (int)((long long)cap0 * g / ng), g1 = (int)((long long)cap0 * (g + 1) / ng);
      if (g1 > g0) s->tsegs.push_back(Seg{nullptr, g0, g1, 0, false, {}});
    }
  if (s->cap[1] > 0) s->tsegs.push_back(Seg{nullptr, cap0, N, 1, true, {}});
  const int nseg = (int)s->tsegs.size();
  tick_free(s); // (an earlier attempt that failed half way left its allocations behind: ADVICE r04)
  int rc = alloc_dev(&s->d_segctl, nseg);
  if (!rc) rc = alloc_pinned(&s->h_segctl, 2 * (size_t)nseg);
  if (!rc) rc = alloc_dev(&s->d_modectl, 2);
  if (!rc) rc = alloc_pinned(&s->h_modectl, 2 * 2);
  if (!rc) rc = alloc_pinned(&s->h_count_word, 4);
  if (!rc) rc = alloc_dev(&s->d_slot_ticket, N);
  if (!rc) rc = alloc_dev(&s->d_admit_idx, N);
  if (rc) return rc;
  // an item list holds at most one evaluation (+ its speculative twin) per slot of the segment
  const int maxpos = 8 * ((max_chunks_upto(s->w * s->h) + 7) / 8);
  if (maxpos >= (1 << kTickChunkBits)) return invalid("dsm_stream (tick engine): a level has too many chunks for the item encoding");
  if (N >= (1 << (30 - kTickChunkBits))) return invalid("dsm_stream (tick engine): too many slots for the item encoding");
  s->d_items.assign(2 * nseg, nullptr);
  s->items_cap.assign(nseg, 0);
  s->last_count.assign(nseg, 0);
  s->last_chain.assign(nseg, 0);
  for (int si = 0; si < nseg; si++) {
    const int ns = s->tsegs[si].i1 - s->tsegs[si].i0;
    const int cap = ns * 2 * maxpos + 64;
    s->items_cap[si] = cap;
    for (int b = 0; b < 2; b++) // item slots [0, cap), then one chain entry per slot of the segment
      if ((rc = alloc_dev(&s->d_items[2 * si + b], (size_t)cap + ns))) return rc;
  }
  DSM_HIP(hipMemsetAsync(s->d_segctl, 0, sizeof(TickSegCtl) * nseg, ctx->stream));
  DSM_HIP(hipMemsetAsync(s->d_slot_ticket, 0, sizeof(unsigned long long) * N, ctx->stream));
  // the waiting ring and the result ring of each kind: the host appends waiting problems and never lets more in than the result
  // ring has room for; the device consumes / produces by monotonic counters
  for (int mode = 0; mode < 2; mode++) {
    int ring = 1024;
    while (ring < 4 * s->cap[mode]) ring <<= 1;
    s->ring[mode] = ring;
    if ((rc = alloc_dev(&s->d_pending[mode], ring)) || (rc = alloc_pinned(&s->h_pending[mode], ring)) || (rc = alloc_dev(&s->d_results[mode], ring)) ||
        (rc = alloc_pinned(&s->h_results[mode], 2 * (size_t)ring)))
      return rc;
    TickModeCtl mc;
    memset(&mc, 0, sizeof mc);
    mc.ring = ring;
    s->h_modectl[mode] = mc;
  }
  DSM_HIP(hipMemcpyAsync(s->d_modectl, s->h_modectl, sizeof(TickModeCtl) * 2, hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < 2; k++) {
    DSM_HIP(hipEventCreate(&s->ev_begin[k]));
    DSM_HIP(hipEventCreate(&s->ev_end[k]));
  }
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  memset(s->seen, 0, sizeof s->seen);
  s->tick_ready = true;
  return DSM_OK;
}

// the read-back of advance `k` (parity k & 1): control blocks and result rings are in the pinned copies once ev_end[k & 1] is reached
static int tick_collect(dsm_stream *s, long long k) {
  dsm_context *ctx = s->ctx;
  const int par = (int)(k & 1), nseg = (int)s->tsegs.size();
  DSM_HIP(hipEventSynchronize(s->ev_end[par]));
  float ms = 0;
  DSM_HIP(hipEventElapsedTime(&ms, s->ev_begin[par], s->ev_end[par]));
  s->stats[0].total_ms += ms, s->stats[1].total_ms += ms; // (the stream's statistics are cumulative)
  s->stats[0].polls += 1;
  if (s->inflight_timed[par]) collect_eval_timing(ctx, s->ev_lvl, 1, s->stats[0]);
  for (int si = 0; si < nseg; si++) {
    const TickSegCtl &sc = s->h_segctl[par * nseg + si];
    if (sc.overflow) {
      set_error("internal: an item list of the tick engine ran over");
      return DSM_ERR_STATE;
    }
    s->last_count[si] = sc.count[s->inflight_parity[par]];
    s->last_chain[si] = sc.chain[s->inflight_parity[par]];
  }
  for (int mode = 0; mode < 2; mode++) {
    const TickModeCtl &mc = s->h_modectl[par * 2 + mode];
    dsm_stream::Seen &sn = s->seen[mode];
    const int ring = s->ring[mode];
    const long long new_ret = mc.retired - sn.retired;
    if (new_ret < 0 || new_ret > ring) {
      set_error("internal: the tick engine's result ring ran over");
      return DSM_ERR_STATE;
    }
    dsm_stats &st = s->stats[mode];
    for (int l = 0; l < s->nlevels; l++) {
      st.evals[l] += mc.sched_evals[l] - sn.evals[l];
      st.evals_residual_only[l] += mc.sched_ro[l] - sn.ro[l];
      sn.evals[l] = mc.sched_evals[l], sn.ro[l] = mc.sched_ro[l];
      st.launches[l] += s->inflight_ticks[par];
    }
    const TickResult *res = s->h_results[mode] + (size_t)par * ring;
    long long life_sum = 0;
    for (long long i = 0; i < new_ret; i++) {
      const TickResult &R = res[(sn.retired + i) & (ring - 1)];
      life_sum += R.ticks; // (= its LM rounds, less the rounds chains ran inside one tick)
      auto it = s->origin.find(R.ticket);
      if (it == s->origin.end()) {
        set_error("internal: the tick engine returned an unknown ticket");
        return DSM_ERR_STATE;
      }
      const dsm_stream::Origin &og = it->second;
      dsm_stream_result r;
      memset(&r, 0, sizeof r);
      r.ticket = R.ticket;
      r.kind = mode;
      r.status = R.status;
      r.passes = (int)(k - og.admitted_at + 1);
      if (mode == 0) {
        const bool wrote = R.status == ST_GOOD || R.status == ST_BAD_AFFINE; // as dsm_track_batch (:612-613 / :598)
        memcpy(r.pose, wrote ? R.cur : og.pose0, sizeof r.pose);
        memcpy(r.aff, wrote ? R.aff_cur : og.aff0, sizeof r.aff);
        r.good = R.status == ST_GOOD ? 1 : 0;
        memcpy(r.flow, R.flow, sizeof r.flow);
        r.scale = 1.0f;
      } else {
        r.good = 1;
        r.scale = R.scale_cur;                  // :954
        r.err = (float)R.last_residuals[0];     // :963
        r.pose[3] = 1.0;
      }
      memcpy(r.last_residuals, R.last_residuals, sizeof r.last_residuals);
      for (int l = 0; l < DSM_MAX_LEVELS; l++) {
        r.evals[l] = R.evals[l];
        // SURVEY.md 8d's bytes, booked when the problem retires (exact per problem; a long run's average equals the bytes moved)
        const long long nl = og.trk->desc.lv[l].n, img = 12ll * (og.trk->w >> l) * (og.trk->h >> l);
        if (l < s->nlevels) st.algorithmic_bytes += R.evals[l] * (16ll * nl + (48ll * nl < img ? 48ll * nl : img));
      }
      s->done.push_back(r);
      s->retired[mode]++;
      s->origin.erase(it);
    }
    if (new_ret > 0 && mode == (s->cap[0] > 0 ? 0 : 1)) { // (a tick = one LM round of every resident problem)
      const double mean = (double)life_sum / (double)new_ret;
      s->life_ticks = s->life_ticks > 0.0 ? 0.75 * s->life_ticks + 0.25 * mean : mean;
    }
    sn.retired = mc.retired;
    sn.admitted = mc.pending_head;
    s->resident[mode] = (int)(sn.admitted - sn.retired);
  }
  return DSM_OK;
}

// everything in flight is read back
static int tick_sync(dsm_stream *s) {
  while (s->collected < s->advances) {
    const int rc = tick_collect(s, s->collected);
    if (rc) return rc;
    s->collected++;
  }
  return DSM_OK;
}

// One advance of the tick engine, PIPELINED: waiting problems are appended to the device's waiting ring, free slots take them,
// then `ticks` ticks -- evaluate every staged item, step every resident problem; finished problems retire into the result ring and
// their slots take the next waiting problem on the spot -- and the control blocks and result rings are copied back behind them.
// The call returns as soon as that is ENQUEUED; what it reads back is the PREVIOUS advance's (its kernels ran while the host
// prepared this one), so results surface one advance late and the device never waits for the host (dsm_stream_sync / drain wait
// for everything).  With dsm_context_set_timing the advance is synchronous (its events are read right away).
static int tick_advance(dsm_stream *s) {
  dsm_context *ctx = s->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  if (!s->partial_stride) return DSM_OK; // nothing was ever submitted
  int rc;
  if (!s->tick_ready && (rc = tick_setup(s))) return rc;
  if (s->collected == s->advances && s->resident[0] + s->resident[1] == 0 && s->waiting[0].empty() && s->waiting[1].empty() &&
      s->handed[0] == s->seen[0].admitted && s->handed[1] == s->seen[1].admitted)
    return DSM_OK; // idle
  const int nseg = (int)s->tsegs.size();
  const int n_track_segs = s->cap[1] > 0 ? nseg - 1 : nseg;
  rc = ensure_streams(ctx, n_track_segs < 1 ? 1 : n_track_segs, s->cap[1] > 0 && n_track_segs > 0);
  if (rc) return rc;
  for (int si = 0; si < nseg; si++)
    s->tsegs[si].st = si == 0 ? ctx->stream : s->tsegs[si].companion ? ctx->companion_stream : ctx->extra_streams[si - 1];
  if ((s->advances - s->collected >= 2 || ctx->timing) && (rc = tick_sync(s))) return rc; // (at most one advance behind: the pinned copies are two deep;
                                                                                          // a timed advance reads its own events alone)
  const long long k = s->advances;
  const int par = (int)(k & 1);
  // ---- waiting problems -> the device's ring: as many as the waiting ring and the result ring have room for ----
  std::vector<dsm_tracker *> fresh;
  int n_new[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++) {
    const int ring = s->ring[mode];
    const long long room = std::min<long long>(ring - (s->handed[mode] - s->seen[mode].admitted), ring - (s->handed[mode] - s->seen[mode].retired));
    const int n = (int)std::min<long long>((long long)s->waiting[mode].size(), room > 0 ? room : 0);
    n_new[mode] = n;
    for (int i = 0; i < n; i++) {
      const Waiting &wt = s->waiting[mode].front();
      TickPending &pd = s->h_pending[mode][(s->handed[mode] + i) & (ring - 1)];
      pd.start = wt.start;
      pd.trk = wt.trk->d_desc;
      pd.ticket = wt.ticket;
      fresh.push_back(wt.trk);
      s->sched_fixed_schedule = wt.trk->params.fixed_schedule, s->sched_speculate = wt.trk->params.speculate, s->have_sched = true;
      auto it = s->origin.find(wt.ticket);
      if (it != s->origin.end()) it->second.admitted_at = k;
      s->waiting[mode].pop_front();
    }
  }
  if (!s->have_sched) return invalid("dsm_stream: internal: no problem was ever handed over");
  if (!fresh.empty() && (rc = sync_descs(ctx, fresh.data(), (int)fresh.size()))) return rc;
  DSM_HIP(hipEventRecord(s->ev_begin[par], ctx->stream));
  if (ctx->timing) DSM_HIP(hipEventRecord(ctx->ev_total[0], ctx->stream)); // (collect_eval_timing places the dispatches relative to it)
  for (int mode = 0; mode < 2; mode++) {
    if (!n_new[mode]) continue;
    const int ring = s->ring[mode];
    const int first = (int)(s->handed[mode] & (ring - 1)), n1 = std::min(n_new[mode], ring - first);
    DSM_HIP(hipMemcpyAsync(s->d_pending[mode] + first, s->h_pending[mode] + first, sizeof(TickPending) * n1, hipMemcpyHostToDevice, ctx->stream));
    if (n_new[mode] > n1)
      DSM_HIP(hipMemcpyAsync(s->d_pending[mode], s->h_pending[mode], sizeof(TickPending) * (n_new[mode] - n1), hipMemcpyHostToDevice, ctx->stream));
    s->handed[mode] += n_new[mode];
    s->h_count_word[par * 2 + mode] = s->handed[mode];
    DSM_HIP(hipMemcpyAsync(&s->d_modectl[mode].pending_count, &s->h_count_word[par * 2 + mode], sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  }
  const bool may_admit[2] = {s->handed[0] > s->seen[0].admitted, s->handed[1] > s->seen[1].admitted}; // (something may be waiting on the device)
  if (may_admit[0] || may_admit[1]) {
    TickReserveArgs ra;
    if (nseg > kTickMaxSegs) return invalid("dsm_stream: too many stream groups");
    ra.nseg = nseg;
    for (int si = 0; si < nseg; si++) ra.seg[si] = TickSegDesc{s->tsegs[si].mode, s->tsegs[si].i0, s->tsegs[si].i1 - s->tsegs[si].i0};
    launch_tick_reserve(ctx->stream, ra, s->d_states, s->d_modectl, s->d_admit_idx);
  }
  if (nseg > 1) {
    DSM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
    for (int si = 1; si < nseg; si++) DSM_HIP(hipStreamWaitEvent(s->tsegs[si].st, ctx->fork_event, 0));
  }
  int T = s->ticks;
  if (s->auto_ticks && s->life_ticks > 0.0) {
    const int m = s->cap[0] > 0 ? 0 : 1;
    // sized to retire 7/8 of what the advance hands over: the rest stays in the device's waiting ring as the buffer from which a slot
    // freed by a retirement is refilled on the spot (a caller that hands over a full pool every advance builds a backlog of an eighth of
    // a pool per advance; it is worked off at full occupancy when the caller stops).  Nothing handed over: the residents' own life.
    // (With dsm_params.fixed_schedule every problem lives the same number of ticks and a cohort admitted together stays on one level,
    // which evaluates in larger single-level launches: there an advance is one cohort's whole life.)
    const double share = s->sched_fixed_schedule > 0 ? 1.0 : 0.875;
    const double t = n_new[m] > 0 ? share * (double)n_new[m] * s->life_ticks / (double)s->cap[m] : s->life_ticks;
    T = (int)std::ceil(t);
    T = T < 8 ? 8 : T > 256 ? 256 : T;
  }
  s->inflight_ticks[par] = T;
  size_t ev_used = 0;
  s->ev_lvl.clear();
  const int speculate = s->sched_fixed_schedule > 0 ? 0 : s->sched_speculate;
  // (with dsm_params.fixed_schedule a cohort moves in lock step and an advance is its whole life: chains would only unbalance it)
  const int chain_rounds = (s->sched_fixed_schedule > 0 || !s->chain_possible) ? 0 : s->chain_rounds;
  for (int si = nseg - 1; si >= 0; si--) {
    const Seg &sg = s->tsegs[si];
    const int i0 = sg.i0, ns = sg.i1 - sg.i0, mode = sg.mode;
    if (may_admit[mode]) // (free slots take what tick_reserve_kernel gave them)
      launch_tick_admit(sg.st, mode, ns, (const TrackerDev **)s->d_tracker_ptrs + i0, s->d_states + i0,
                        TickList{s->d_items[2 * si + s->parity], s->d_segctl + si, s->parity, s->items_cap[si], chain_rounds > 0 ? ns : 0, s->chain_max_n0},
                        s->d_modectl + mode, s->d_pending[mode], s->d_slot_ticket + i0, s->d_admit_idx + i0);
  }
  for (int t = 0; t < T; t++) {
    const int buf = (s->parity + t) & 1;
    for (int si = nseg - 1; si >= 0; si--) {
      const Seg &sg = s->tsegs[si];
      const int i0 = sg.i0, ns = sg.i1 - sg.i0, mode = sg.mode;
      // grid: what the segment's list held at the end of the last advance read back, plus slack (the kernel strides over a longer list)
      long long grid = (long long)(s->last_count[si] + s->last_chain[si]) * 5 / 4 + 256;
      if (s->last_count[si] + s->last_chain[si] == 0) grid = (long long)ns * 16;
      if (grid > s->items_cap[si]) grid = s->items_cap[si];
      float *part = s->d_partials + (size_t)i0 * s->partial_stride;
      hipEvent_t ea = nullptr, eb = nullptr;
      if (ctx->timing && !sg.companion) {
        ea = get_event(ctx, ev_used++);
        eb = get_event(ctx, ev_used++);
        s->ev_lvl.push_back(0); // (all levels in one launch: booked under index 0)
        if (ea) DSM_HIP(hipEventRecord(ea, sg.st));
      }
      const TickList next{s->d_items[2 * si + (buf ^ 1)], s->d_segctl + si, buf ^ 1, s->items_cap[si], chain_rounds > 0 ? ns : 0, s->chain_max_n0};
      const TickChainArgs chain{(const TrackerDev **)s->d_tracker_ptrs + i0, s->d_states + i0, next, s->d_modectl + mode, s->d_pending[mode], s->d_results[mode],
                                s->d_slot_ticket + i0, chain_rounds, s->chain_flags};
      launch_tick_eval(sg.st, mode, (int)grid, s->d_states + i0, part, s->partial_stride, s->d_items[2 * si + buf], s->d_segctl + si, buf, s->items_cap[si], chain, s->chain_possible);
      if (eb) DSM_HIP(hipEventRecord(eb, sg.st));
      launch_tick_lm(sg.st, mode, ns, (const TrackerDev **)s->d_tracker_ptrs + i0, s->d_states + i0, part, s->partial_stride, next, s->d_modectl + mode,
                     s->d_pending[mode], s->d_results[mode], s->d_slot_ticket + i0, speculate, s->lm_opts);
    }
  }
  s->parity = (s->parity + T) & 1;
  s->inflight_parity[par] = s->parity;
  s->inflight_timed[par] = ctx->timing;
  DSM_HIP(hipGetLastError());
  for (int si = 1; si < nseg; si++) {
    hipEvent_t ev = s->tsegs[si].companion ? ctx->companion_event : ctx->join_events[si - 1];
    DSM_HIP(hipEventRecord(ev, s->tsegs[si].st));
    DSM_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
  }
  // ---- read-back behind the ticks: control blocks and the result rings, into this advance's pinned copies ----
  DSM_HIP(hipMemcpyAsync(s->h_modectl + par * 2, s->d_modectl, sizeof(TickModeCtl) * 2, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipMemcpyAsync(s->h_segctl + par * nseg, s->d_segctl, sizeof(TickSegCtl) * nseg, hipMemcpyDeviceToHost, ctx->stream));
  for (int mode = 0; mode < 2; mode++)
    if (s->handed[mode] > s->seen[mode].retired) // (results may appear)
      DSM_HIP(hipMemcpyAsync(s->h_results[mode] + (size_t)par * s->ring[mode], s->d_results[mode], sizeof(TickResult) * s->ring[mode], hipMemcpyDeviceToHost,
                             ctx->stream));
  DSM_HIP(hipEventRecord(s->ev_end[par], ctx->stream));
  s->advances++;
  s->total_ticks += T;
  s->passes++;
  // what was in flight before this advance has finished on the device by now, or finishes while the host is busy here
  if (ctx->timing || !s->pipelined) return tick_sync(s);
  while (s->collected < s->advances - 1) {
    rc = tick_collect(s, s->collected);
    if (rc) return rc;
    s->collected++;
  }
  return DSM_OK;
}

extern "C" {

int dsm_stream_sync(dsm_stream *s) {
  if (!s) return invalid("null stream");
  if (s->engine != 1 || !s->tick_ready) return DSM_OK;
  DSM_HIP(hipSetDevice(s->ctx->device));
  return tick_sync(s);
}

int dsm_stream_set_pipelined(dsm_stream *s, int on) {
  if (!s) return invalid("null stream");
  const int rc = dsm_stream_sync(s);
  if (rc) return rc;
  s->pipelined = on != 0;
  return DSM_OK;
}

int dsm_stream_drain(dsm_stream *s) {
  if (!s) return invalid("null stream");
  long long last_retired = -1;
  int idle_advances = 0;
  for (;;) {
    int rc = dsm_stream_sync(s); // (the tail is short chains: nothing to overlap)
    const long long ret_now = s->retired[0] + s->retired[1];
    idle_advances = ret_now == last_retired ? idle_advances + 1 : 0;
    last_retired = ret_now;
    if (idle_advances > 4096) { // (a problem's whole life is a few dozen ticks)
      set_error("internal: dsm_stream_drain makes no progress");
      return DSM_ERR_STATE;
    }
    if (rc) return rc;
    int resident = 0, waiting = 0;
    dsm_stream_counts(s, &resident, &waiting, nullptr);
    if (resident == 0 && waiting == 0) {
      if (lm_spin_expired() != 0) { // (a wave hand-shake inside an LM step gave up waiting: results cannot be trusted -- never seen; the waits are bounded so that a defect shows here, not as a hung device)
        set_error("internal: a bounded wait inside an LM step expired");
        return DSM_ERR_STATE;
      }
      return DSM_OK;
    }
    rc = dsm_stream_advance(s);
    if (rc) return rc;
  }
}

} // extern "C"
