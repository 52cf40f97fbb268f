// reference_binding_check.cpp -- instantiates dsm_host::ReferenceTracker (ReferenceBinding.hpp: the bodies behind the public
// methods of the reference's dso::TrackerAndScaler) with minimal stand-ins that carry exactly the member names the binding
// touches on Sophus::SE3 / dso::AffLight / Eigen vectors / dso::FrameHessian / dso::CalibHessian, and drives it the way
// FrontEnd.cpp does (makeK, setCoarseTrackingRef, trackNewestCoarse, optimizeScale) on the fixture of
// tests/test_host_adaptor.py::test_reference_binding_through_standin_types_equals_the_plain_adaptor (same format as host_adaptor_demo.cpp).  Prints the same JSON line as host_adaptor_demo, so
// the test can require the two to be equal bit for bit: what is tested is the binding's conversion code, not the stand-ins.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ReferenceBinding.hpp"

namespace standin {
struct Quat { // Eigen::Quaterniond as returned by Sophus::SE3d::unit_quaternion()
  double c[4]; // x, y, z, w
  double x() const { return c[0]; }
  double y() const { return c[1]; }
  double z() const { return c[2]; }
  double w() const { return c[3]; }
};
template <int N, class T = double>
struct Vec { // Eigen::Matrix<T, N, 1>
  T v[N];
  Vec() {
    for (int i = 0; i < N; i++) v[i] = T(0);
  }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  T *data() { return v; }
  const T *data() const { return v; }
};
struct SE3 { // Sophus::SE3d
  Quat q;
  Vec<3> t;
  SE3() {
    q.c[0] = q.c[1] = q.c[2] = 0, q.c[3] = 1;
  }
  const Quat &unit_quaternion() const { return q; }
  const Vec<3> &translation() const { return t; }
};
struct AffLight { // dso::AffLight
  double a, b;
  AffLight(double a_ = 0, double b_ = 0) : a(a_), b(b_) {}
};
struct Mat33f { // Eigen::Matrix3f
  float m[9];
  float operator()(int r, int c) const { return m[3 * r + c]; }
};
struct FrameShell { // dso::FrameShell
  int id, incoming_id;
};
struct FrameHessian { // dso::FrameHessian: the members the tracker reads
  Vec<3, float> *dIp[DSM_MAX_LEVELS]; // Eigen::Vector3f* per level
  float ab_exposure;
  FrameShell *shell;
  AffLight aff;
  AffLight aff_g2l() const { return aff; }
};
struct CalibHessian { // dso::CalibHessian
  float k[4];
  float fxl() const { return k[0]; }
  float fyl() const { return k[1]; }
  float cxl() const { return k[2]; }
  float cyl() const { return k[3]; }
};
static int g_levels = 0; // dso::pyrLevelsUsed

struct Types { // the traits struct a maintainer writes for the real types
  typedef standin::SE3 SE3;
  typedef standin::AffLight AffLight;
  typedef standin::Vec<5> Vec5;
  typedef standin::Vec<3> Vec3;
  typedef standin::FrameHessian FrameHessian;
  typedef standin::CalibHessian CalibHessian;
  typedef standin::Mat33f Mat33f;
  static int levels() { return g_levels; }
  static SE3 make_se3(const double q[4], const double t[3]) {
    SE3 T;
    for (int i = 0; i < 4; i++) T.q.c[i] = q[i];
    for (int i = 0; i < 3; i++) T.t[i] = t[i];
    return T;
  }
  static void fill_params(dsm_params &) {} // the fixture runs on the library's defaults (= the recalled upstream settings)
};
} // namespace standin

template <typename T>
static void rd(FILE *f, T *p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s fixture.bin\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int w, h, nl;
  rd(f, &w, 1), rd(f, &h, 1), rd(f, &nl, 1);
  standin::g_levels = nl;
  standin::CalibHessian calib;
  rd(f, calib.k, 4);
  std::vector<double> T(16);
  rd(f, T.data(), 16);
  std::vector<std::vector<float>> u(nl), v(nl), id(nl), c(nl);
  dsm_host::TemplateLists tpl;
  for (int l = 0; l < nl; l++) {
    int n;
    rd(f, &n, 1);
    u[l].resize(n), v[l].resize(n), id[l].resize(n), c[l].resize(n);
    rd(f, u[l].data(), n), rd(f, v[l].data(), n), rd(f, id[l].data(), n), rd(f, c[l].data(), n);
    tpl.n[l] = n, tpl.pc_u[l] = u[l].data(), tpl.pc_v[l] = v[l].data(), tpl.pc_idepth[l] = id[l].data(), tpl.pc_color[l] = c[l].data();
  }
  typedef standin::Vec<3, float> Texel;
  static_assert(sizeof(Texel) == 12, "packed float3, as Eigen::Vector3f");
  std::vector<std::vector<Texel>> newp(nl), rightp(nl);
  for (auto *pyr : {&newp, &rightp})
    for (int l = 0; l < nl; l++) {
      (*pyr)[l].resize((size_t)(w >> l) * (h >> l));
      rd(f, (float *)(*pyr)[l].data(), 3 * (*pyr)[l].size());
    }
  fclose(f);

  dsm_context *ctx = nullptr;
  if (dsm_context_create(0, &ctx) != DSM_OK) {
    fprintf(stderr, "no device: %s\n", dsm_last_error());
    return 3;
  }
  {
    standin::FrameShell ref_shell = {7, 100}, new_shell = {8, 101}, right_shell = {8, 102};
    standin::FrameHessian ref_fh, new_fh, right_fh;
    for (int l = 0; l < nl; l++) ref_fh.dIp[l] = newp[l].data(), new_fh.dIp[l] = newp[l].data(), right_fh.dIp[l] = rightp[l].data();
    ref_fh.ab_exposure = new_fh.ab_exposure = right_fh.ab_exposure = 1.0f;
    ref_fh.shell = &ref_shell, new_fh.shell = &new_shell, right_fh.shell = &right_shell;

    standin::Mat33f K1 = {{calib.k[0], 0, calib.k[2], 0, calib.k[1], calib.k[3], 0, 0, 1}};
    dsm_host::ReferenceTracker<standin::Types> tracker(ctx, w, h, T, K1); // FrontEnd.cpp:57-58
    tracker.makeK(&calib);                                                 // :797
    tracker.setCoarseTrackingRef({&ref_fh}, tpl);                          // :798
    standin::SE3 lastF_2_fh;                                               // identity
    standin::AffLight aff_g2l(0, 0);
    standin::Vec<5> minRes, achieved;
    for (int i = 0; i < 5; i++) minRes[i] = NAN;
    const bool good = tracker.trackNewestCoarse(&new_fh, lastF_2_fh, aff_g2l, nl - 1, minRes, achieved); // :204-206
    float scale = 1.0f;
    const float err = tracker.optimizeScale(&right_fh, scale, nl - 1); // :992
    const auto q = lastF_2_fh.unit_quaternion();
    printf("{\"good\": %d, \"pose\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"aff\": [%.17g, %.17g], "
           "\"last0\": %.9g, \"flow\": [%.9g, %.9g, %.9g], \"scale\": %.9g, \"scale_err\": %.9g, \"ref_id\": %d}\n",
           good ? 1 : 0, q.x(), q.y(), q.z(), q.w(), lastF_2_fh.translation()[0], lastF_2_fh.translation()[1], lastF_2_fh.translation()[2],
           aff_g2l.a, aff_g2l.b, achieved[0], tracker.lastFlowIndicators[0], tracker.lastFlowIndicators[1], tracker.lastFlowIndicators[2],
           scale, err, tracker.refFrameID);
  }
  dsm_context_destroy(ctx);
  return 0;
}
