// host_adaptor_demo.cpp -- exercises dsm_host::TrackerAndScaler (the C++ adaptor with the reference's
// class surface) end to end from a binary fixture written by tests/test_host_adaptor.py:
//   int32 w,h,nl ; float K[4] ; double T[16] ; per level: int32 n, then u,v,idepth,color (n floats each) ;
//   per level new-frame dIp (3*w_l*h_l floats) ; per level right-frame dIp.
// Prints one JSON line with the tracked pose, affine parameters, residuals and the optimised scale.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "TrackerAndScaler.hpp"

template <typename T>
static void rd(FILE *f, T *p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s fixture.bin [chunk_geometry]\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int w, h, nl;
  rd(f, &w, 1);
  rd(f, &h, 1);
  rd(f, &nl, 1);
  float K[4];
  rd(f, K, 4);
  std::vector<double> T(16);
  rd(f, T.data(), 16);
  std::vector<std::vector<float>> u(nl), v(nl), id(nl), c(nl), newp(nl), rightp(nl);
  dsm_host::TemplateLists tpl;
  for (int l = 0; l < nl; l++) {
    int n;
    rd(f, &n, 1);
    u[l].resize(n), v[l].resize(n), id[l].resize(n), c[l].resize(n);
    rd(f, u[l].data(), n), rd(f, v[l].data(), n), rd(f, id[l].data(), n), rd(f, c[l].data(), n);
    tpl.n[l] = n;
    tpl.pc_u[l] = u[l].data(), tpl.pc_v[l] = v[l].data(), tpl.pc_idepth[l] = id[l].data(), tpl.pc_color[l] = c[l].data();
  }
  std::vector<const float *> newptr(nl), rightptr(nl);
  for (int l = 0; l < nl; l++) {
    newp[l].resize(3 * (size_t)(w >> l) * (h >> l));
    rd(f, newp[l].data(), newp[l].size());
    newptr[l] = newp[l].data();
  }
  for (int l = 0; l < nl; l++) {
    rightp[l].resize(3 * (size_t)(w >> l) * (h >> l));
    rd(f, rightp[l].data(), rightp[l].size());
    rightptr[l] = rightp[l].data();
  }
  fclose(f);

  dsm_context *ctx = nullptr;
  if (dsm_context_create(0, &ctx) != DSM_OK) {
    fprintf(stderr, "no device: %s\n", dsm_last_error());
    return 3;
  }
  {
    dsm_params prm; // library defaults unless the caller names the reduction geometry (the reference binding uses the latency table)
    dsm_host::check(DSM_PARAMS_INIT(&prm), "DSM_PARAMS_INIT");
    if (argc > 2) prm.chunk_geometry = atoi(argv[2]);
    dsm_host::TrackerAndScaler tracker(ctx, w, h, nl, T, K, &prm);
    tracker.makeK(K[0], K[1], K[2], K[3]);
    dsm_host::FrameView ref, nf, rf;
    ref.shell_id = 7;
    nf.dIp = newptr.data(), nf.unique_id = 1;
    rf.dIp = rightptr.data(), rf.unique_id = 2;
    tracker.setCoarseTrackingRef(ref, tpl);
    dsm_host::SE3 pose;
    dsm_host::AffLight aff;
    double minres[5] = {NAN, NAN, NAN, NAN, NAN}, last[5];
    const bool good = tracker.trackNewestCoarse(nf, pose, aff, nl - 1, minres, last);
    float scale = 1.0f;
    const float err = tracker.optimizeScale(rf, scale, nl - 1);
    // the same two problems through the streaming form (dsm_host::Stream): frames are resident from the calls above
    int stream_equal = 0;
    {
      dsm_host::Stream stream(ctx, 2, 2);
      const uint64_t tk = stream.submitTrack(tracker, dsm_host::SE3(), dsm_host::AffLight(), nl - 1, minres);
      const uint64_t ts = stream.submitScale(tracker, 1.0f, nl - 1);
      stream.drain();
      std::vector<dsm_stream_result> res;
      stream.results(res);
      for (const dsm_stream_result &r : res) {
        if (r.ticket == tk) {
          bool same = (r.good != 0) == good && r.aff[0] == aff.a && r.aff[1] == aff.b;
          for (int i = 0; i < 4; i++) same = same && r.pose[i] == pose.q[i];
          for (int i = 0; i < 3; i++) same = same && r.pose[4 + i] == pose.t[i];
          stream_equal += same;
        } else if (r.ticket == ts) {
          stream_equal += r.scale == scale && r.err == err;
        }
      }
    }
    printf("{\"good\": %d, \"pose\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g], \"aff\": [%.17g, %.17g], "
           "\"last0\": %.9g, \"flow\": [%.9g, %.9g, %.9g], \"scale\": %.9g, \"scale_err\": %.9g, \"ref_id\": %d, \"stream_results_equal\": %d}\n",
           good ? 1 : 0, pose.q[0], pose.q[1], pose.q[2], pose.q[3], pose.t[0], pose.t[1], pose.t[2], aff.a, aff.b, last[0],
           tracker.lastFlowIndicators[0], tracker.lastFlowIndicators[1], tracker.lastFlowIndicators[2], scale, err,
           tracker.refFrameID, stream_equal);
  }
  dsm_context_destroy(ctx);
  return 0;
}
