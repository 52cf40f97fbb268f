// ReferenceBinding.hpp -- the reference-side wrapper as compilable code: the bodies a maintainer puts behind the public methods of
// dso::TrackerAndScaler (src/scale_optimization/TrackerAndScaler.h:38-64) so that FrontEnd.cpp's call sites (:204-206,
// :797-798, :992-998, :1032) compile unchanged while the work runs on the C ABI.
//
// It is a class TEMPLATE over the reference's own types, named by a traits struct, because none of them (Sophus::SE3,
// dso::AffLight, Eigen Vec5 / Mat33f, dso::FrameHessian, dso::CalibHessian) exists in this repository's build image.  The
// conversions only use members both the real types and any stand-in with the same member names provide:
//   SE3           T.unit_quaternion().x() .y() .z() .w(),  T.translation()[i]      (Sophus::SE3d)
//   AffLight      .a, .b, AffLight(a, b)                                            (dso::AffLight, util/NumType.h upstream)
//   Vec5          v[i]                                                              (Eigen::Matrix<double,5,1>)
//   Vec3          v[i]                                                              (lastFlowIndicators, TrackerAndScaler.h:63)
//   Mat33f        K(r, c)
//   FrameHessian  fh->dIp[lvl][0].data() (Eigen::Vector3f[] is packed float3, read at TrackerAndScaler.cpp:709,1016),
//                 fh->ab_exposure (:718), fh->shell->id (:323), fh->shell->incoming_id, fh->aff_g2l() (:324)
//   CalibHessian  HCalib->fxl() fyl() cxl() cyl()                                   (:121-124)
// and three things the traits struct supplies because they are globals or constructors on the reference side:
//   static int levels();                                             // dso::pyrLevelsUsed
//   static SE3 make_se3(const double q_xyzw[4], const double t[3]);  // SE3(Eigen::Quaterniond(w,x,y,z), Vec3(t))
//   static void fill_params(dsm_params &p);                          // setting_huberTH, setting_coarseCutoffTH, SCALE_*, affine modes
// INTEGRATION.md section 1 shows the ten-line traits struct for the real types.  tests/test_host_adaptor.py::test_reference_binding_through_standin_types_equals_the_plain_adaptor instantiates
// the template with minimal stand-ins of those member names (host/reference_binding_check.cpp) and checks that a track + scale
// optimisation through it returns, bit for bit, what the plain adaptor returns: the conversions are tested code, not prose.
#pragma once
#include <memory>
#include <vector>

#include "TrackerAndScaler.hpp"

namespace dsm_host {

template <class R>
class ReferenceTracker {
public:
  typedef typename R::SE3 SE3r;
  typedef typename R::AffLight AffLightr;
  typedef typename R::Vec5 Vec5r;
  typedef typename R::Vec3 Vec3r;
  typedef typename R::FrameHessian FrameHessianr;
  typedef typename R::CalibHessian CalibHessianr;
  typedef typename R::Mat33f Mat33fr;

  // TrackerAndScaler(int w, int h, const std::vector<double>& tfm_vec, const Mat33f& K1)   (TrackerAndScaler.cpp:47-109)
  ReferenceTracker(dsm_context *ctx, int ww, int hh, const std::vector<double> &tfm_vec, const Mat33fr &K1)
      : lastRef(nullptr), refFrameID(-1), lastRef_aff_g2l(0, 0), firstCoarseRMSE(-1) {
    dsm_params p;
    if (DSM_PARAMS_INIT(&p) != DSM_OK) throw std::runtime_error(dsm_last_error()); // (the size this host was compiled with)
    R::fill_params(p);
    // the reference tracks ONE frame at a time (FrontEnd.cpp:585-686): short chunks on the large levels (an evaluation through sooner), the
    // levels of at most 4096 points as a chain -- their LM loop (:505-593) in one launch instead of one launch per round
    p.chunk_geometry = 2;
    p.persistent_coarse = -1;
    const float k1[4] = {K1(0, 0), K1(1, 1), K1(0, 2), K1(1, 2)}; // fx1_, fy1_, cx1_, cy1_ (:89-98)
    impl_.reset(new TrackerAndScaler(ctx, ww, hh, R::levels(), tfm_vec, k1, &p));
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = 1000; // :460
  }

  // void makeK(CalibHessian* HCalib)   (:117-141)
  void makeK(CalibHessianr *HCalib) { impl_->makeK(HCalib->fxl(), HCalib->fyl(), HCalib->cxl(), HCalib->cyl()); }

  // void setCoarseTrackingRef(std::vector<FrameHessian*> frameHessians)   (:317-327): the reference's own makeCoarseDepthL0
  // (:143-315, it walks the PointHessian graph) has run and filled pc_u_/pc_v_/pc_idepth_/pc_color_/pc_n_, handed over as tpl
  void setCoarseTrackingRef(const std::vector<FrameHessianr *> &frameHessians, const TemplateLists &tpl) {
    lastRef = frameHessians.back(); // :319
    ref_view_ = view(lastRef, ref_keep_);
    impl_->setCoarseTrackingRef(ref_view_, tpl);
    refFrameID = impl_->refFrameID;           // :323
    lastRef_aff_g2l = lastRef->aff_g2l();     // :324
    firstCoarseRMSE = impl_->firstCoarseRMSE; // :326
  }

  // void scaleCoarseDepthL0(float scale)   (:329-336)
  void scaleCoarseDepthL0(float scale) { impl_->scaleCoarseDepthL0(scale); }

  // bool trackNewestCoarse(FrameHessian* newFrameHessian, SE3& lastToNew_out, AffLight& aff_g2l_out, int coarsestLvl,
  //                        Vec5 minResForAbort, Vec5& lastResiduals, IOWrap::Output3DWrapper* wrap = 0)   (:451-638)
  bool trackNewestCoarse(FrameHessianr *newFrameHessian, SE3r &lastToNew_out, AffLightr &aff_g2l_out, int coarsestLvl, Vec5r minResForAbort,
                         Vec5r &lastResiduals, void * /*wrap*/ = nullptr) {
    const FrameView v = view(newFrameHessian, new_keep_);
    SE3 T = to_dsm(lastToNew_out);
    AffLight a(aff_g2l_out.a, aff_g2l_out.b);
    double mr[5], lr[5];
    for (int i = 0; i < 5; i++) mr[i] = minResForAbort[i];
    const bool ok = impl_->trackNewestCoarse(v, T, a, coarsestLvl, mr, lr);
    lastToNew_out = R::make_se3(T.q, T.t);
    aff_g2l_out = AffLightr(a.a, a.b);
    for (int i = 0; i < 5; i++) lastResiduals[i] = lr[i];
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = impl_->lastFlowIndicators[i]; // :597
    return ok;
  }

  // float optimizeScale(FrameHessian* fh1, float& scale, int coarsestLvl)   (:854-964)
  float optimizeScale(FrameHessianr *fh1, float &scale, int coarsestLvl) {
    const FrameView v = view(fh1, right_keep_);
    return impl_->optimizeScale(v, scale, coarsestLvl);
  }

  // act as pure output (TrackerAndScaler.h:59-64)
  FrameHessianr *lastRef;
  int refFrameID;
  AffLightr lastRef_aff_g2l;
  Vec3r lastFlowIndicators;
  double firstCoarseRMSE;

  TrackerAndScaler &adaptor() { return *impl_; }

  // dso::FrameHessian -> FrameView (the fields read at TrackerAndScaler.cpp:709,718,1016,323,324); `keep` owns the level pointers
  static FrameView view(const FrameHessianr *fh, std::vector<const float *> &keep) {
    FrameView v;
    keep.resize(R::levels());
    for (int l = 0; l < R::levels(); l++) keep[l] = fh->dIp[l][0].data();
    v.dIp = keep.data();
    v.ab_exposure = fh->ab_exposure;
    v.shell_id = fh->shell->id;
    v.aff_g2l = AffLight(fh->aff_g2l().a, fh->aff_g2l().b);
    v.unique_id = fh->shell->incoming_id; // skips the re-upload across the 5 + 26 k hypotheses of FrontEnd::trackNewCoarse
    return v;
  }
  // Sophus::SE3 -> quaternion (Eigen coefficient order x, y, z, w) + translation
  static SE3 to_dsm(const SE3r &T) {
    SE3 o;
    const auto q = T.unit_quaternion();
    o.q[0] = q.x(), o.q[1] = q.y(), o.q[2] = q.z(), o.q[3] = q.w();
    for (int i = 0; i < 3; i++) o.t[i] = T.translation()[i];
    return o;
  }

private:
  std::unique_ptr<TrackerAndScaler> impl_;
  FrameView ref_view_;
  std::vector<const float *> ref_keep_, new_keep_, right_keep_;
};

} // namespace dsm_host
