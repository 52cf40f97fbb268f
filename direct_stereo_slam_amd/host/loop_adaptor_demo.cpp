// loop_adaptor_demo.cpp -- exercises the C++ loop-detection adaptor (LoopDetection.hpp: the surface of search_place.h,
// ScanContext.h and generate_spherical_points.h on the C ABI) the way LoopHandler::run does (LoopHandler.cpp:186-262):
// per keyframe generate_spherical_points -> ScanContext::generate -> search_ringkey -> search_sc.  Reads a binary fixture
// written by tests/test_host_adaptor.py:
//   int32 n_frames ; double lidar_range ; per frame: int32 n_kf, n_pts ; kf ids (n_kf int32) ; kf poses (6 n_kf doubles) ;
//   cur_cw (12 doubles) ; point kf ids (n_pts int32) ; points (3 n_pts doubles)
// Prints one JSON line per frame: selected count, ring key, candidates, best match and difference.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "LoopDetection.hpp"

template <typename T>
static void rd(FILE *f, T *p, size_t n) {
  if (n && fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s fixture.bin trajectory_out.txt\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int n_frames;
  double lidar_range;
  rd(f, &n_frames, 1);
  rd(f, &lidar_range, 1);
  dsm_context *ctx = nullptr;
  if (dsm_context_create(0, &ctx) != DSM_OK) {
    fprintf(stderr, "no device: %s\n", dsm_last_error());
    return 3;
  }
  {
    dsm_host::ScanContext sc; // 60 x 20
    dsm_host::RingKeyIndex ringkeys(ctx, (int)sc.getHeight());
    std::vector<dsm_host::SigType> signatures; // loop_frames_[i]->signature
    std::vector<int> ids;
    std::vector<double> traj;
    for (int fr = 0; fr < n_frames; fr++) {
      int n_kf, n_pts;
      rd(f, &n_kf, 1);
      rd(f, &n_pts, 1);
      std::vector<int> kf(n_kf), pk(n_pts);
      std::vector<double> poses(6 * (size_t)n_kf), cw(12), xyz(3 * (size_t)n_pts);
      rd(f, kf.data(), n_kf), rd(f, poses.data(), poses.size()), rd(f, cw.data(), 12), rd(f, pk.data(), n_pts), rd(f, xyz.data(), xyz.size());
      std::vector<std::pair<int, std::vector<double>>> id_pose_wc, pts_nearby;
      for (int k = 0; k < n_kf; k++) id_pose_wc.push_back({kf[k], std::vector<double>(poses.begin() + 6 * k, poses.begin() + 6 * k + 6)});
      for (int i = 0; i < n_pts; i++) pts_nearby.push_back({pk[i], std::vector<double>(xyz.begin() + 3 * i, xyz.begin() + 3 * i + 3)});
      std::vector<double> pts_spherical;
      dsm_host::generate_spherical_points(pts_nearby, id_pose_wc, cw.data(), lidar_range, pts_spherical); // LoopHandler.cpp:186-187
      std::vector<float> ringkey;
      dsm_host::SigType signature;
      double tfm[16];
      sc.generate(pts_spherical, ringkey, signature, lidar_range, tfm); // :236
      std::vector<int> candidates;
      ringkeys.search_ringkey(ringkey.data(), candidates); // :247
      int matched = -1;
      float diff = -1.f;
      if (!candidates.empty())
        dsm_host::search_sc(signature, [&](int i) -> const dsm_host::SigType & { return signatures[i]; }, candidates, (int)sc.getWidth(), matched,
                            diff); // :256
      signatures.push_back(signature);
      ids.push_back(fr);
      traj.push_back(-cw[3]), traj.push_back(-cw[7]), traj.push_back(-cw[11]);
      printf("{\"frame\": %d, \"n_sel\": %d, \"n_kf_kept\": %d, \"ringkey\": [", fr, (int)(pts_spherical.size() / 3), (int)id_pose_wc.size());
      for (size_t i = 0; i < ringkey.size(); i++) printf("%s%.9g", i ? ", " : "", ringkey[i]);
      printf("], \"n_sig\": %d, \"candidates\": [", (int)signature.size());
      for (size_t i = 0; i < candidates.size(); i++) printf("%s%d", i ? ", " : "", candidates[i]);
      printf("], \"matched\": %d, \"diff\": %.9g, \"index_size\": %d}\n", matched, diff, (int)ringkeys.size());
    }
    dsm_host::save_trajectory(argv[2], ids, traj); // LoopHandler.cpp:59-80
  }
  fclose(f);
  dsm_context_destroy(ctx);
  return 0;
}
