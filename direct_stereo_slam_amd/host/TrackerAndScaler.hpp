// TrackerAndScaler.hpp -- C++ host adaptor that keeps the public surface of the reference's
// dso::TrackerAndScaler (src/scale_optimization/TrackerAndScaler.h:38-64) on top of the C ABI
// (include/dsm_hotpath.h), so FrontEnd.cpp's call sites (FrontEnd.cpp:204-206, :797-798,
// :992-998, :1032) keep their shape.  Header-only, plain C++11, no Eigen/Sophus/DSO needed:
// the DSO types are reduced to the fields this path actually reads.  INTEGRATION.md shows the
// three-line conversions from the real dso::FrameHessian / Sophus::SE3 / dso::AffLight.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dsm_hotpath.h"

namespace dsm_host {

// Sophus::SE3d as the path uses it: unit quaternion (Eigen coefficient order x,y,z,w) + translation
struct SE3 {
  double q[4] = {0, 0, 0, 1};
  double t[3] = {0, 0, 0};
};
// dso::AffLight
struct AffLight {
  double a = 0, b = 0;
  AffLight() {}
  AffLight(double a_, double b_) : a(a_), b(b_) {}
};
// The fields of dso::FrameHessian read by the tracker: dIp[lvl] (TrackerAndScaler.cpp:709,1016),
// ab_exposure (:718), shell->id (:323), aff_g2l() (:324)
struct FrameView {
  const float *const *dIp = nullptr; // per level AoS (I,dx,dy), w_l*h_l texels
  float ab_exposure = 1.0f;
  int shell_id = -1;
  AffLight aff_g2l;
  long long unique_id = -1; // identity of the pixel data, to skip repeated uploads
};
// Output of makeCoarseDepthL0 (TrackerAndScaler.cpp:143-315): pc_u/pc_v/pc_idepth/pc_color + pc_n
struct TemplateLists {
  int n[DSM_MAX_LEVELS] = {0};
  const float *pc_u[DSM_MAX_LEVELS] = {nullptr};
  const float *pc_v[DSM_MAX_LEVELS] = {nullptr};
  const float *pc_idepth[DSM_MAX_LEVELS] = {nullptr};
  const float *pc_color[DSM_MAX_LEVELS] = {nullptr};
};

inline void check(int rc, const char *what) {
  if (rc != DSM_OK) throw std::runtime_error(std::string(what) + ": " + dsm_last_error());
}
// the structures of dsm_hotpath.h this translation unit was compiled with must be the library's
inline void check_abi() {
  if (dsm_abi_version() != DSM_ABI_VERSION)
    throw std::runtime_error("libdsm_hotpath implements ABI version " + std::to_string(dsm_abi_version()) + ", this host was built against " +
                             std::to_string(DSM_ABI_VERSION));
}

class TrackerAndScaler {
public:
  // reference: TrackerAndScaler(int w, int h, const std::vector<double>& tfm_vec, const Mat33f& K1)
  TrackerAndScaler(dsm_context *ctx, int w, int h, int pyrLevelsUsed, const std::vector<double> &tfm_vec,
                   const float K1_fx_fy_cx_cy[4], const dsm_params *params = nullptr)
      : refFrameID(-1), lastRef(nullptr), lastRef_aff_g2l(0, 0), firstCoarseRMSE(-1), ctx_(ctx), levels_(pyrLevelsUsed) {
    if (tfm_vec.size() != 16) throw std::invalid_argument("tfm_vec must hold 16 doubles");
    check_abi();
    check(dsm_tracker_create(ctx, w, h, pyrLevelsUsed, tfm_vec.data(), K1_fx_fy_cx_cy, params, &t_), "dsm_tracker_create");
    lastFlowIndicators[0] = lastFlowIndicators[1] = lastFlowIndicators[2] = 1000;
  }
  ~TrackerAndScaler() { dsm_tracker_destroy(t_); }
  TrackerAndScaler(const TrackerAndScaler &) = delete;
  TrackerAndScaler &operator=(const TrackerAndScaler &) = delete;

  // reference: makeK(CalibHessian*) reads fxl(), fyl(), cxl(), cyl() (TrackerAndScaler.cpp:121-124)
  void makeK(float fxl, float fyl, float cxl, float cyl) { check(dsm_tracker_make_k(t_, fxl, fyl, cxl, cyl), "makeK"); }

  // reference: setCoarseTrackingRef(std::vector<FrameHessian*>) -- the host keeps makeCoarseDepthL0
  // (it walks the PointHessian graph) and hands its result over
  void setCoarseTrackingRef(const FrameView &ref, const TemplateLists &tpl) {
    check(dsm_tracker_set_ref(t_, ref.shell_id, ref.aff_g2l.a, ref.aff_g2l.b, ref.ab_exposure, tpl.n, tpl.pc_u, tpl.pc_v,
                              tpl.pc_idepth, tpl.pc_color),
          "setCoarseTrackingRef");
    lastRef = &ref;
    refFrameID = ref.shell_id;     // :323
    lastRef_aff_g2l = ref.aff_g2l; // :324
    firstCoarseRMSE = -1;          // :326
  }

  // The same, with makeCoarseDepthL0 (TrackerAndScaler.cpp:143-315) on the device: the caller flattens the window's
  // active points (centerProjectedTo and the weight of :155-158, what the loops at :149-164 read from the
  // PointHessian graph).  `frameOwner` is the tracker whose NEW_LEFT slot still holds the keyframe's pyramid (the
  // one that tracked it; the reference swaps its two trackers, FrontEnd.cpp:627-632).
  void setCoarseTrackingRef(const FrameView &ref, TrackerAndScaler &frameOwner, int npts, const float *pu, const float *pv,
                            const float *pidepth, const float *pweight, int *pc_n_out = nullptr) {
    check(dsm_tracker_set_ref_from_points(t_, frameOwner.t_, DSM_SLOT_NEW_LEFT, ref.shell_id, ref.aff_g2l.a, ref.aff_g2l.b,
                                          ref.ab_exposure, npts, pu, pv, pidepth, pweight, pc_n_out),
          "setCoarseTrackingRef (device template)");
    lastRef = &ref;
    refFrameID = ref.shell_id;
    lastRef_aff_g2l = ref.aff_g2l;
    firstCoarseRMSE = -1;
  }

  void scaleCoarseDepthL0(float scale) { check(dsm_tracker_scale_depth(t_, scale), "scaleCoarseDepthL0"); }

  // reference: bool trackNewestCoarse(FrameHessian*, SE3&, AffLight&, int, Vec5, Vec5&, Output3DWrapper*)
  bool trackNewestCoarse(const FrameView &newFrameHessian, SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl,
                         const double minResForAbort[5], double lastResiduals[5]) {
    upload(newFrameHessian, DSM_SLOT_NEW_LEFT);
    double pose[7] = {lastToNew_out.q[0], lastToNew_out.q[1], lastToNew_out.q[2], lastToNew_out.q[3],
                      lastToNew_out.t[0], lastToNew_out.t[1], lastToNew_out.t[2]};
    double aff[2] = {aff_g2l_out.a, aff_g2l_out.b};
    double mr[DSM_MAX_LEVELS], lr[DSM_MAX_LEVELS];
    for (int i = 0; i < DSM_MAX_LEVELS; i++) mr[i] = (i < 5 && minResForAbort) ? minResForAbort[i] : NAN;
    int good = 0;
    check(dsm_tracker_track(t_, pose, aff, coarsestLvl, mr, lr, lastFlowIndicators, &good), "trackNewestCoarse");
    for (int i = 0; i < 4; i++) lastToNew_out.q[i] = pose[i];
    for (int i = 0; i < 3; i++) lastToNew_out.t[i] = pose[4 + i];
    aff_g2l_out.a = aff[0];
    aff_g2l_out.b = aff[1];
    if (lastResiduals)
      for (int i = 0; i < 5; i++) lastResiduals[i] = lr[i];
    return good != 0;
  }

  // reference: float optimizeScale(FrameHessian* fh1, float& scale, int coarsestLvl)
  float optimizeScale(const FrameView &fh1, float &scale, int coarsestLvl) {
    upload(fh1, DSM_SLOT_NEW_RIGHT);
    float err = NAN;
    check(dsm_tracker_optimize_scale(t_, &scale, coarsestLvl, &err), "optimizeScale");
    return err;
  }

  // the untrapped branch of FrontEnd::optimizeScale (FrontEnd.cpp:995-1003) as ONE batched call: optimizeScale from every
  // initial guess, the smallest positive error wins (the first on ties); returns scale_error, new_scale by reference
  float optimizeScaleGuesses(const FrameView &fh1, const std::vector<float> &scale_guess, float &new_scale, int coarsestLvl) {
    upload(fh1, DSM_SLOT_NEW_RIGHT);
    float err = -1.0f;
    check(dsm_tracker_optimize_scale_guesses(t_, (int)scale_guess.size(), scale_guess.data(), coarsestLvl, &new_scale, &err, nullptr, nullptr),
          "optimizeScaleGuesses");
    return err;
  }

  dsm_tracker *handle() { return t_; }

  // "next" row N1: hand over the camera image itself (DSM_PIXEL_F32: the undistorted float image FrontEnd.cpp:605,680 feed to
  // makeImages; DSM_PIXEL_U8: the "mono8" camera bytes, main.cpp:216-217) instead of a host-built (I,dx,dy) pyramid: the pyramid
  // is built on the device.  A FrameView carrying the same unique_id afterwards counts as resident in that slot (its dIp may be
  // null), so trackNewestCoarse / optimizeScale skip their upload.  row_pitch_bytes: 0 = tight rows.
  void uploadImage(int slot, const void *pixels, int pixel_type, float ab_exposure, long long unique_id, size_t row_pitch_bytes = 0) {
    dsm_tracker *ts[1] = {t_};
    const int slots[1] = {slot};
    const void *imgs[1] = {pixels};
    const float ex[1] = {ab_exposure};
    check(dsm_upload_images(ctx_, 1, ts, slots, imgs, ex, pixel_type, row_pitch_bytes), "uploadImage");
    uploaded_[slot] = unique_id;
  }

  // (uploadImages below: the pixels with this identity were handed over to `slot` by a batched call)
  void noteResident(int slot, long long unique_id) { uploaded_[slot] = unique_id; }

  // act as pure output (TrackerAndScaler.h:59-64)
  int refFrameID;
  const FrameView *lastRef;
  AffLight lastRef_aff_g2l;
  double lastFlowIndicators[3];
  double firstCoarseRMSE;

private:
  void upload(const FrameView &f, int slot) {
    if (f.unique_id >= 0 && f.unique_id == uploaded_[slot]) return; // same pixels already resident
    check(dsm_tracker_upload_frame(t_, slot, f.dIp, f.ab_exposure), "upload_frame");
    uploaded_[slot] = f.unique_id;
  }
  dsm_tracker *t_ = nullptr;
  dsm_context *ctx_ = nullptr;
  int levels_;
  long long uploaded_[2] = {-1, -1};
};

// setCoarseTrackingRef (device template) for the new keyframes of several sequences in ONE call (dsm_set_refs_from_points: the
// jobs' launches back to back, one host synchronisation)
struct RefRequest {
  TrackerAndScaler *tracker;
  const FrameView *ref;         // must outlive the tracker's use of lastRef, as in the single call
  TrackerAndScaler *frameOwner; // whose NEW_LEFT slot holds the keyframe's pyramid
  int npts;
  const float *pu, *pv, *pidepth, *pweight;
};
inline void setCoarseTrackingRefs(dsm_context *ctx, const std::vector<RefRequest> &reqs) {
  if (reqs.empty()) return;
  std::vector<dsm_ref_job> jobs(reqs.size());
  for (size_t i = 0; i < reqs.size(); i++) {
    const RefRequest &q = reqs[i];
    dsm_ref_job &J = jobs[i];
    J.t = q.tracker->handle(), J.frame_owner = q.frameOwner->handle(), J.slot = DSM_SLOT_NEW_LEFT, J.ref_frame_id = q.ref->shell_id;
    J.ref_aff_a = q.ref->aff_g2l.a, J.ref_aff_b = q.ref->aff_g2l.b, J.ref_exposure = q.ref->ab_exposure, J.npts = q.npts;
    J.pu = q.pu, J.pv = q.pv, J.pidepth = q.pidepth, J.pweight = q.pweight, J.n_out = nullptr;
  }
  check(dsm_set_refs_from_points(ctx, (int)jobs.size(), jobs.data()), "setCoarseTrackingRefs (device templates)");
  for (const RefRequest &q : reqs) {
    q.tracker->lastRef = q.ref;
    q.tracker->refFrameID = q.ref->shell_id;     // :323
    q.tracker->lastRef_aff_g2l = q.ref->aff_g2l; // :324
    q.tracker->firstCoarseRMSE = -1;             // :326
  }
}

// uploadImage for many trackers in ONE hand-over (dsm_upload_images: one staging copy, one pyramid launch sequence, one host
// synchronisation for all of them) -- what a node serving several sequences does with the frames that arrived together
// enqueue = true: dsm_upload_images_enqueue -- stream-ordered, no host wait; for page-locked capture buffers that stay untouched until
// the next hand-over (a frame ring)
inline void uploadImages(dsm_context *ctx, const std::vector<TrackerAndScaler *> &trackers, const std::vector<int> &slots,
                         const std::vector<const void *> &pixels, int pixel_type, const std::vector<float> &ab_exposures,
                         const std::vector<long long> &unique_ids, size_t row_pitch_bytes = 0, bool enqueue = false) {
  const size_t n = trackers.size();
  if (!n) return;
  if (slots.size() != n || pixels.size() != n || ab_exposures.size() != n || unique_ids.size() != n) throw std::runtime_error("uploadImages: sizes differ");
  std::vector<dsm_tracker *> ts(n);
  for (size_t i = 0; i < n; i++) ts[i] = trackers[i]->handle();
  if (enqueue)
    check(dsm_upload_images_enqueue(ctx, (int)n, ts.data(), slots.data(), pixels.data(), ab_exposures.data(), pixel_type, row_pitch_bytes), "uploadImages");
  else
    check(dsm_upload_images(ctx, (int)n, ts.data(), slots.data(), pixels.data(), ab_exposures.data(), pixel_type, row_pitch_bytes), "uploadImages");
  for (size_t i = 0; i < n; i++) trackers[i]->noteResident(slots[i], unique_ids[i]);
}

// ---- streaming form of many sequences (dsm_stream_*): continuous admission ------------------------------
// Many TrackerAndScaler instances (one per sequence / hypothesis) share the GPU: each submits its trackNewestCoarse /
// optimizeScale problem and polls for the result; the stream keeps `track_slots` + `scale_slots` problems resident, advances
// them one LM round per tick whatever level each stands on, and refills a slot the moment its problem retires.  Results are
// bit-identical to the trackers' own calls.  The tracker's template and frames must stay untouched until its result is back.
class Stream {
public:
  Stream(dsm_context *ctx, int track_slots, int scale_slots) {
    check_abi();
    check(dsm_stream_create(ctx, track_slots, scale_slots, &s_), "dsm_stream_create");
  }
  ~Stream() { dsm_stream_destroy(s_); }
  Stream(const Stream &) = delete;
  Stream &operator=(const Stream &) = delete;

  // trackNewestCoarse of `tracker` on the frame resident in its NEW_LEFT slot (upload it first: uploadImage, or a FrameView
  // through trackNewestCoarse's own path); returns the ticket the result carries
  uint64_t submitTrack(TrackerAndScaler &tracker, const SE3 &lastToNew_guess, const AffLight &aff_g2l, int coarsestLvl,
                       const double minResForAbort[5] = nullptr) {
    dsm_tracker *t = tracker.handle();
    const double pose[7] = {lastToNew_guess.q[0], lastToNew_guess.q[1], lastToNew_guess.q[2], lastToNew_guess.q[3],
                            lastToNew_guess.t[0], lastToNew_guess.t[1], lastToNew_guess.t[2]};
    const double aff[2] = {aff_g2l.a, aff_g2l.b};
    double mr[DSM_MAX_LEVELS];
    for (int i = 0; i < DSM_MAX_LEVELS; i++) mr[i] = (i < 5 && minResForAbort) ? minResForAbort[i] : NAN;
    uint64_t ticket = 0;
    check(dsm_stream_submit_track(s_, 1, &t, pose, aff, coarsestLvl, mr, &ticket), "dsm_stream_submit_track");
    return ticket;
  }
  // optimizeScale of `tracker` on the frame resident in its NEW_RIGHT slot
  uint64_t submitScale(TrackerAndScaler &tracker, float scale, int coarsestLvl) {
    dsm_tracker *t = tracker.handle();
    uint64_t ticket = 0;
    check(dsm_stream_submit_scale(s_, 1, &t, &scale, coarsestLvl, &ticket), "dsm_stream_submit_scale");
    return ticket;
  }
  void advance() { check(dsm_stream_advance(s_), "dsm_stream_advance"); }
  // LM rounds a problem whose pending evaluation is one chunk may run inside one tick (dsm_stream_set_chain; 0 off, -1 the default)
  void setChain(int max_rounds) { check(dsm_stream_set_chain(s_, max_rounds), "dsm_stream_set_chain"); }
  void drain() { check(dsm_stream_drain(s_), "dsm_stream_drain"); }
  // appends the problems retired so far; dsm_stream_result::pose / aff / good / last_residuals / flow are trackNewestCoarse's
  // outputs, scale / err optimizeScale's
  void results(std::vector<dsm_stream_result> &out) {
    int ready = 0;
    check(dsm_stream_counts(s_, nullptr, nullptr, &ready), "dsm_stream_counts");
    if (ready <= 0) return;
    const size_t at = out.size();
    out.resize(at + (size_t)ready);
    int n = 0;
    check(dsm_stream_results(s_, ready, out.data() + at, &n), "dsm_stream_results");
    out.resize(at + (size_t)n);
  }
  dsm_stream *handle() { return s_; }

private:
  dsm_stream *s_ = nullptr;
};

// ---- "next" row N4: the hypothesis loop of FrontEnd::trackNewCoarse (FrontEnd.cpp:194-256) -----------
// Same results as the reference's sequential loop; try 0 runs alone, the remaining tries as ONE batched
// launch sequence without abort, and the abort / take-over logic (TrackerAndScaler.cpp:598,
// FrontEnd.cpp:225-247) is replayed on the host from their per-level residuals.
struct HypothesesResult {
  bool haveOneGood = false;
  SE3 lastF_2_fh;
  AffLight aff_g2l;
  double flowVecs[3] = {0, 0, 0};
  double achievedRes[5];
  int triesUsed = 0;
};

inline HypothesesResult trackHypotheses(dsm_context *ctx, TrackerAndScaler &tracker, const FrameView &fh,
                                        const std::vector<SE3> &lastF_2_fh_tries, const AffLight &aff_last_2_l,
                                        int coarsestLvl, double last_coarse_rmse0, double reTrackThreshold = 1.5) {
  HypothesesResult R;
  for (double &a : R.achievedRes) a = NAN;
  double flow[3] = {100, 100, 100};
  auto consume = [&](bool good, const SE3 &pose, const AffLight &aff, const double *cur, const double *fl) {
    if (good && std::isfinite((float)cur[0]) && !(cur[0] >= R.achievedRes[0])) { // FrontEnd.cpp:225-233
      for (int k = 0; k < 3; k++) flow[k] = fl[k];
      R.aff_g2l = aff;
      R.lastF_2_fh = pose;
      R.haveOneGood = true;
    }
    if (R.haveOneGood) // :236-243
      for (int l = 0; l < 5; l++)
        if (!std::isfinite((float)R.achievedRes[l]) || R.achievedRes[l] > cur[l]) R.achievedRes[l] = cur[l];
    return R.haveOneGood && R.achievedRes[0] < last_coarse_rmse0 * reTrackThreshold; // :245-247
  };
  const size_t n = lastF_2_fh_tries.size();
  bool done = false;
  {
    SE3 pose = lastF_2_fh_tries[0];
    AffLight aff = aff_last_2_l;
    double cur[5];
    const bool good = tracker.trackNewestCoarse(fh, pose, aff, coarsestLvl, R.achievedRes, cur);
    R.triesUsed = 1;
    done = consume(good, pose, aff, cur, tracker.lastFlowIndicators);
  }
  if (!done && n > 1) {
    const int m = (int)n - 1;
    std::vector<dsm_tracker *> ts(m, tracker.handle());
    std::vector<double> poses(7 * m), affs(2 * m), last(DSM_MAX_LEVELS * m), fl(3 * m);
    std::vector<int> good(m);
    for (int i = 0; i < m; i++) {
      const SE3 &T = lastF_2_fh_tries[i + 1];
      for (int k = 0; k < 4; k++) poses[7 * i + k] = T.q[k];
      for (int k = 0; k < 3; k++) poses[7 * i + 4 + k] = T.t[k];
      affs[2 * i] = aff_last_2_l.a, affs[2 * i + 1] = aff_last_2_l.b;
    }
    check(dsm_track_batch(ctx, m, ts.data(), poses.data(), affs.data(), coarsestLvl, nullptr, last.data(), fl.data(),
                          good.data()),
          "dsm_track_batch");
    for (int i = 0; i < m && !done; i++) {
      R.triesUsed++;
      double cur[5];
      for (int l = 0; l < 5; l++) cur[l] = last[DSM_MAX_LEVELS * i + l];
      bool g = good[i] != 0;
      for (int l = coarsestLvl; l >= 0; l--)
        if (cur[l] > 1.5 * R.achievedRes[l]) { // the abort the sequential run would have taken (:598)
          for (int k = 0; k < l; k++) cur[k] = NAN;
          g = false;
          break;
        }
      SE3 pose = lastF_2_fh_tries[i + 1];
      if (g) {
        for (int k = 0; k < 4; k++) pose.q[k] = poses[7 * i + k];
        for (int k = 0; k < 3; k++) pose.t[k] = poses[7 * i + 4 + k];
      }
      done = consume(g, pose, AffLight(affs[2 * i], affs[2 * i + 1]), cur, &fl[3 * i]);
    }
  }
  if (!R.haveOneGood) { // :249-256
    R.lastF_2_fh = lastF_2_fh_tries[0];
    R.aff_g2l = aff_last_2_l;
    flow[0] = flow[1] = flow[2] = 0;
  }
  for (int k = 0; k < 3; k++) R.flowVecs[k] = flow[k];
  return R;
}

// ---- "next" row N2: dso::PoseEstimator (PoseEstimator.h:34-83) on the C ABI ---------------------------
class PoseEstimator {
public:
  PoseEstimator(dsm_context *ctx, int w, int h, int pyrLevelsUsed, const dsm_params *params = nullptr) {
    check_abi();
    check(dsm_pose_estimator_create(ctx, w, h, pyrLevelsUsed, params, &pe_), "dsm_pose_estimator_create");
  }
  ~PoseEstimator() { dsm_pose_estimator_destroy(pe_); }
  PoseEstimator(const PoseEstimator &) = delete;
  PoseEstimator &operator=(const PoseEstimator &) = delete;

  // reference: bool estimate(const std::vector<std::pair<Eigen::Vector3d, float*>>& pts, float ref_ab_exposure,
  //                          FrameHessian* new_fh, const std::vector<float>& new_cam, int coarsest_lvl,
  //                          Eigen::Matrix4d& ref_to_new, float& pose_error)
  // pts: xyz as n x 3 doubles, ref_colors[lvl][i] = pts[i].second[lvl]; ref_to_new: row-major 4x4
  bool estimate(int n, const double *xyz, const float *const *ref_colors, float ref_ab_exposure, const FrameView &new_fh,
                const float new_cam[4], int coarsest_lvl, double ref_to_new[16], float &pose_error) {
    int ok = 0;
    check(dsm_pose_estimator_estimate(pe_, n, xyz, ref_colors, ref_ab_exposure, new_fh.dIp, new_fh.ab_exposure, new_cam,
                                      coarsest_lvl, ref_to_new, &pose_error, &ok),
          "PoseEstimator::estimate");
    return ok != 0;
  }

private:
  dsm_pose_estimator *pe_ = nullptr;
};

} // namespace dsm_host
