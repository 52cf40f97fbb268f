// LoopDetection.hpp -- C++ host adaptor for the loop-detection half of the hot path: the surface of
// src/loop_closure/loop_detection/{search_place.h, ScanContext.h, generate_spherical_points.h} and of the flann index
// LoopHandler owns (LoopHandler.cpp:35-39), on top of the C ABI (include/dsm_hotpath.h).  Same names, argument meaning
// and error behaviour as the reference; no Eigen / FLANN types at the boundary: points are rows of three doubles,
// ring keys are float arrays, SigType is the reference's own typedef.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dsm_hotpath.h"

namespace dsm_host {

#ifndef DSM_HOST_CHECK
#define DSM_HOST_CHECK
inline void loop_check(int rc, const char *what) {
  if (rc != DSM_OK) throw std::runtime_error(std::string(what) + ": " + dsm_last_error());
}
#endif

constexpr int kFlannNN = 3;          // FLANN_NN      search_place.h:21
constexpr int kLoopMargin = 100;     // LOOP_MARGIN   search_place.h:22
constexpr float kRingkeyThres = 0.1f; // RINGKEY_THRES search_place.h:23

typedef std::vector<std::pair<int, double>> SigType; // ScanContext.h:24

// replaces `flann::Index<flann::L2<float>> *ringkeys_` (LoopHandler.cpp:35-39) together with the function-static delay
// queue of search_ringkey (search_place.h:43-45): per-object state instead of statics.  shard_rank / shard_count: this
// process holds the ordinals `ordinal mod shard_count == shard_rank` of the index (one process per GPU); attach a
// communicator and search_ringkey becomes a collective call that returns the same candidates on every rank.
class RingKeyIndex {
public:
  RingKeyIndex(dsm_context *ctx, int ringkey_dim, const float *dummy_key = nullptr, int shard_rank = 0, int shard_count = 1)
      : dim_(ringkey_dim) {
    if (dsm_abi_version() != DSM_ABI_VERSION) throw std::runtime_error("libdsm_hotpath: ABI version mismatch with this host's dsm_hotpath.h");
    loop_check(dsm_ringdb_create(ctx, ringkey_dim, kLoopMargin, kFlannNN, kRingkeyThres, dummy_key, 4096, shard_rank, shard_count, &db_),
               "dsm_ringdb_create");
  }
  ~RingKeyIndex() { dsm_ringdb_destroy(db_); }
  RingKeyIndex(const RingKeyIndex &) = delete;
  RingKeyIndex &operator=(const RingKeyIndex &) = delete;

  void attach(dsm_comm *comm) { loop_check(dsm_ringdb_attach_comm(db_, comm), "dsm_ringdb_attach_comm"); }
  size_t size() const { return (size_t)dsm_ringdb_size(db_); } // flann Index::size(), the dummy row included

  // reference: search_ringkey(const flann::Matrix<float>& ringkey, flann::Index<...>* ringkeys, std::vector<int>& candidates)
  // (search_place.h:25-57): candidates are APPENDED, ordinals of searched frames minus the dummy, nearest first
  void search_ringkey(const float *ringkey, std::vector<int> &candidates) {
    int cand[4], n = 0;
    loop_check(dsm_ringdb_query_then_enqueue(db_, ringkey, cand, &n), "search_ringkey");
    for (int i = 0; i < n; i++) candidates.emplace_back(cand[i]);
  }
  dsm_ringdb *handle() { return db_; }

private:
  dsm_ringdb *db_ = nullptr;
  int dim_;
};

// reference: search_sc(SigType& signature, const std::vector<dso::LoopFrame*>& loop_frames, const std::vector<int>& candidates,
//                      int sc_width, int& res_idx, float& res_diff)  (search_place.h:59-84).  `signature_of(i)` returns
// loop_frames[i]->signature: the caller keeps its LoopFrame type.
template <class SignatureOf>
inline void search_sc(const SigType &signature, SignatureOf signature_of, const std::vector<int> &candidates, int sc_width, int &res_idx,
                      float &res_diff) {
  std::vector<int> a_idx(signature.size());
  std::vector<double> a_val(signature.size());
  for (size_t i = 0; i < signature.size(); i++) a_idx[i] = signature[i].first, a_val[i] = signature[i].second;
  res_idx = candidates[0]; // :63-64
  res_diff = 1.1f;
  std::vector<int> b_idx;
  std::vector<double> b_val;
  for (int c : candidates) {
    const SigType &s = signature_of(c);
    b_idx.resize(s.size()), b_val.resize(s.size());
    for (size_t i = 0; i < s.size(); i++) b_idx[i] = s[i].first, b_val[i] = s[i].second;
    const float cur = dsm_sc_distance(a_idx.data(), a_val.data(), (int)a_idx.size(), b_idx.data(), b_val.data(), (int)b_idx.size(), sc_width);
    if (res_diff > cur) { // :80-83: the first minimal candidate wins
      res_idx = c;
      res_diff = cur;
    }
  }
}

// reference: class ScanContext (ScanContext.h:26-45)
class ScanContext {
public:
  ScanContext() : num_s_(60), num_r_(20) {}            // ScanContext.cpp:68-71
  ScanContext(int s, int r) : num_s_(s), num_r_(r) {}  // :73-76
  unsigned int getHeight() const { return num_r_; }
  unsigned int getWidth() const { return num_s_; }
  // reference: generate(pts_spherical, ringkey, signature, lidar_range, tfm_pca_rig) (ScanContext.cpp:78-141).
  // pts_spherical: n rows of (x, y, z); ringkey: getHeight() floats; tfm_pca_rig: row-major 4x4
  void generate(const std::vector<double> &pts_spherical_xyz, std::vector<float> &ringkey, SigType &signature, double lidar_range,
                double tfm_pca_rig[16]) const {
    const int n = (int)(pts_spherical_xyz.size() / 3);
    ringkey.assign(num_r_, 0.0f);
    std::vector<int> idx((size_t)num_s_ * num_r_);
    std::vector<double> val((size_t)num_s_ * num_r_);
    int ns = 0;
    loop_check(dsm_scancontext_generate(pts_spherical_xyz.data(), n, lidar_range, num_s_, num_r_, ringkey.data(), idx.data(), val.data(), &ns,
                                        tfm_pca_rig),
               "ScanContext::generate");
    for (int i = 0; i < ns; i++) signature.push_back({idx[i], val[i]}); // :126 appends
  }

private:
  int num_s_, num_r_;
};

// reference: generate_spherical_points(pts_nearby, id_pose_wc, cur_cw, lidar_range, pts_spherical)
// (generate_spherical_points.h:27-85) in flat form.  pts_nearby = (keyframe id, world point) pairs, updated in place to the
// selected points as the reference does (:78-84); id_pose_wc = (keyframe id, 6-vector se(3) log) pairs, trimmed in place
// (:33-41); cur_cw: row-major 3x4.  Output order: ascending voxel index (the reference's unordered_map order is unspecified).
inline void generate_spherical_points(std::vector<std::pair<int, std::vector<double>>> &pts_nearby,
                                      std::vector<std::pair<int, std::vector<double>>> &id_pose_wc, const double cur_cw[12],
                                      double lidar_range, std::vector<double> &pts_spherical_xyz) {
  const int n_kf = (int)id_pose_wc.size(), n_pts = (int)pts_nearby.size();
  std::vector<int> kf_ids(n_kf), keep(n_kf ? n_kf : 1), pt_kf(n_pts ? n_pts : 1), sel(n_pts ? n_pts : 1);
  std::vector<double> poses((size_t)6 * (n_kf ? n_kf : 1)), xyz((size_t)3 * (n_pts ? n_pts : 1)), out((size_t)3 * (n_pts ? n_pts : 1));
  for (int k = 0; k < n_kf; k++) {
    kf_ids[k] = id_pose_wc[k].first;
    for (int j = 0; j < 6; j++) poses[6 * k + j] = id_pose_wc[k].second[j];
  }
  for (int i = 0; i < n_pts; i++) {
    pt_kf[i] = pts_nearby[i].first;
    for (int j = 0; j < 3; j++) xyz[3 * i + j] = pts_nearby[i].second[j];
  }
  int n_out = 0;
  loop_check(dsm_generate_spherical_points(n_kf, kf_ids.data(), poses.data(), cur_cw, lidar_range, n_pts, pt_kf.data(), xyz.data(), keep.data(),
                                           &n_out, sel.data(), out.data()),
             "generate_spherical_points");
  std::vector<std::pair<int, std::vector<double>>> kept_kf, new_pts;
  for (int k = 0; k < n_kf; k++)
    if (keep[k]) kept_kf.push_back(id_pose_wc[k]);
  id_pose_wc.swap(kept_kf);
  for (int i = 0; i < n_out; i++) {
    pts_spherical_xyz.insert(pts_spherical_xyz.end(), out.begin() + 3 * i, out.begin() + 3 * i + 3); // :79 push_back
    new_pts.push_back(pts_nearby[sel[i]]);                                                             // :80
  }
  pts_nearby.swap(new_pts); // :84
}

// reference: the file output of LoopHandler::savePose (LoopHandler.cpp:59-80)
inline void save_trajectory(const char *path, const std::vector<int> &incoming_ids, const std::vector<double> &t_wc_xyz) {
  loop_check(dsm_write_trajectory(path, (int)incoming_ids.size(), incoming_ids.data(), t_wc_xyz.data()), "save_trajectory");
}

} // namespace dsm_host
