// host_capi.cpp -- the pieces of the path that SURVEY.md section 8a keeps on the host by design:
// search_sc (<= 3 candidates per query, src/loop_closure/loop_detection/search_place.h:59-84).
// Plain C++; no device code.
#include "../../include/dsm_hotpath.h"
#include <algorithm>
#include <cstdio>

extern "C" {

// LoopHandler::savePose, LoopHandler.cpp:59-80: one line per loop frame, "incoming_id x y z", std::setprecision(6) on a default
// (%g-style) stream -- the dslam.txt / sodso.txt trajectory files the evaluation scripts of the reference read
int dsm_write_trajectory(const char *path, int n, const int *incoming_ids, const double *t_wc) {
  if (!path || n < 0 || (n && (!incoming_ids || !t_wc))) return DSM_ERR_INVALID;
  FILE *f = fopen(path, "w");
  if (!f) return DSM_ERR_STATE;
  for (int i = 0; i < n; i++)
    fprintf(f, "%d %.6g %.6g %.6g\n", incoming_ids[i], t_wc[3 * i], t_wc[3 * i + 1], t_wc[3 * i + 2]);
  return fclose(f) == 0 ? DSM_OK : DSM_ERR_STATE;
}

// inner loop of search_sc, search_place.h:67-79: float accumulator, double products
float dsm_sc_distance(const int *a_idx, const double *a_val, int na, const int *b_idx, const double *b_val, int nb,
                      int sc_width) {
  float cur_prod = 0;
  int m = 0, n = 0;
  while (m < na && n < nb) {
    if (a_idx[m] == b_idx[n]) {
      cur_prod += a_val[m] * b_val[n];
      m++;
      n++;
    } else {
      if (a_idx[m] < b_idx[n])
        m++;
      else
        n++;
    }
  }
  const float cur_diff = (1 - cur_prod / sc_width) / 2.0;
  return cur_diff;
}

// search_sc, search_place.h:59-84: first minimal candidate wins (strict <)
int dsm_search_sc(const int *sig_idx, const double *sig_val, int n_sig, int n_cand, const int *cand_ids,
                  const int *const *cand_idx, const double *const *cand_val, const int *cand_n, int sc_width,
                  int *res_idx, float *res_diff) {
  if (n_cand < 1 || !cand_ids || !res_idx || !res_diff) return DSM_ERR_INVALID;
  *res_idx = cand_ids[0];
  *res_diff = 1.1;
  for (int c = 0; c < n_cand; c++) {
    const float cur = dsm_sc_distance(sig_idx, sig_val, n_sig, cand_idx[c], cand_val[c], cand_n[c], sc_width);
    if (*res_diff > cur) {
      *res_idx = cand_ids[c];
      *res_diff = cur;
    }
  }
  return DSM_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// ScanContext::generate, ScanContext.cpp:78-141 (+ align_points_PCA :19-66)
// ---------------------------------------------------------------------------------------------
#include <cmath>
#include <cstring>
#include <vector>

#include "loopdet_internal.hpp"

namespace dsm {

} // namespace dsm
using dsm::eig3_sym;

extern "C" {

int dsm_scancontext_generate(const double *pts, int n, double lidar_range, int num_s, int num_r, float *ringkey,
                             int *sig_idx, double *sig_val, int *n_sig_out, double *tfm) {
  if (!pts || n < 1 || num_s < 1 || num_r < 1 || !ringkey || !sig_idx || !sig_val || !n_sig_out || !tfm)
    return DSM_ERR_INVALID;
  // align_points_PCA :19-66
  double mx = 0, my = 0, mz = 0;
  for (int i = 0; i < n; i++) {
    mx += pts[3 * i];
    my += pts[3 * i + 1];
    mz += pts[3 * i + 2];
  }
  mx /= n;
  my /= n;
  mz /= n;
  // the covariance (:40): kCovLanes interleaved partial sums, added in ascending order (loopdet_internal.hpp -- the device form's order)
  std::vector<double> part(6 * (size_t)dsm::kCovLanes, 0.0); // [xx xy xz yy yz zz][lane]
  for (int i = 0; i < n; i++) {
    const double x = pts[3 * i] - mx, y = pts[3 * i + 1] - my, z = pts[3 * i + 2] - mz;
    const int l = i % dsm::kCovLanes;
    part[0 * dsm::kCovLanes + l] += x * x, part[1 * dsm::kCovLanes + l] += x * y, part[2 * dsm::kCovLanes + l] += x * z;
    part[3 * dsm::kCovLanes + l] += y * y, part[4 * dsm::kCovLanes + l] += y * z, part[5 * dsm::kCovLanes + l] += z * z;
  }
  double m6[6];
  for (int k = 0; k < 6; k++) {
    double acc = part[(size_t)k * dsm::kCovLanes];
    for (int l = 1; l < dsm::kCovLanes; l++) acc += part[(size_t)k * dsm::kCovLanes + l];
    m6[k] = acc;
  }
  double cov[9] = {0};
  cov[0] = m6[0], cov[1] = m6[1], cov[2] = m6[2], cov[4] = m6[3], cov[5] = m6[4], cov[8] = m6[5];
  cov[3] = cov[1], cov[6] = cov[2], cov[7] = cov[5];
  double ev[3], V[9];
  eig3_sym(cov, ev, V);
  for (int i = 0; i < 16; i++) tfm[i] = (i % 5 == 0) ? 1.0 : 0.0; // :55-64
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) tfm[r * 4 + c] = V[c * 3 + r]; // row r = v_r^T
  for (int r = 0; r < 3; r++) tfm[r * 4 + 3] = -(tfm[r * 4 + 0] * mx + tfm[r * 4 + 1] * my + tfm[r * 4 + 2] * mz);

  // generate :78-141
  for (int i = 0; i < num_r; i++) ringkey[i] = 0.0f;
  std::vector<double> max_height((size_t)num_s * num_r, -lidar_range - 1.0);
  for (int i = 0; i < n; i++) {
    const double x = pts[3 * i] - mx, y = pts[3 * i + 1] - my, z = pts[3 * i + 2] - mz;
    const double xp = x * V[0] + y * V[3] + z * V[6]; // pts_mat * v0  (x: up)
    const double yp = x * V[1] + y * V[4] + z * V[7];
    const double zp = x * V[2] + y * V[5] + z * V[8];
    const double rho = std::sqrt(yp * yp + zp * zp);
    double theta = std::atan2(zp, yp);
    while (theta < 0) theta += 2.0 * M_PI;
    while (theta >= 2.0 * M_PI) theta -= 2.0 * M_PI;
    const int si = theta / (2.0 * M_PI) * num_s;
    const int ri = rho / lidar_range * num_r;
    if (ri >= num_r) continue; // :113-114
    if (si >= num_s) continue; // the reference only asserts this (:112, compiled out); never write out of bounds
    double &mh = max_height[(size_t)si * num_r + ri];
    mh = std::max(mh, xp);
  }
  std::vector<double> norm(num_s, 0.0);
  int ns = 0;
  for (int i = 0; i < num_s * num_r; i++)
    if (max_height[i] >= -lidar_range) { // :124-131
      ringkey[i % num_r] += 1.0f;
      sig_idx[ns] = i;
      sig_val[ns] = max_height[i];
      ns++;
      norm[i / num_r] += max_height[i] * max_height[i];
    }
  for (int i = 0; i < num_r; i++) ringkey[i] /= num_s; // :134-136
  for (int s = 0; s < num_s; s++) norm[s] = std::sqrt(norm[s]);
  for (int k = 0; k < ns; k++) sig_val[k] /= norm[sig_idx[k] / num_r]; // :139-141
  *n_sig_out = ns;
  return DSM_OK;
}

// ---------------------------------------------------------------------------------------------
// generate_spherical_points, src/loop_closure/loop_detection/generate_spherical_points.h:27-85 (flat-array form)
// ---------------------------------------------------------------------------------------------
} // extern "C"
namespace dsm {
// Sophus SO3::exp as a rotation matrix (Rodrigues; the series below 1e-10 as Sophus does for the quaternion)
void so3_exp_matrix(const double w[3], double R[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = std::sqrt(th2);
  double a, b; // R = I + a W + b W^2
  if (th < 1e-10) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
  } else {
    a = std::sin(th) / th;
    b = (1.0 - std::cos(th)) / th2;
  }
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double w2 = 0;
      for (int k = 0; k < 3; k++) w2 += W[i * 3 + k] * W[k * 3 + j];
      R[i * 3 + j] = (i == j ? 1.0 : 0.0) + a * W[i * 3 + j] + b * w2;
    }
}
// |SO3::log(R)|: the angle of the shortest rotation, from the unit quaternion as Sophus does (2 atan(|v| / w))
double rotation_angle(const double R[9]) {
  // Eigen quaternion-from-matrix, the branch with the largest pivot
  const double tr = R[0] + R[4] + R[8];
  double q[4]; // x y z w
  if (tr > 0) {
    double t = std::sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t, q[1] = (R[2] - R[6]) * t, q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n < 1e-10) return 0.0;
  if (std::fabs(q[3]) < 1e-10) return M_PI;
  return std::fabs(2.0 * std::atan(n / q[3]));
}

// generate_spherical_points.h:33-41: keyframes whose orientation differs from the current one by more than 0.5 rad are trimmed
void trim_keyframes(int n_kf, const double *kf_pose_wc, const double *cur_cw, int *kf_keep) {
  for (int k = 0; k < n_kf; k++) {
    double Rk[9], Rd[9];
    so3_exp_matrix(kf_pose_wc + 6 * k + 3, Rk);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rd[i * 3 + j] = cur_cw[i * 4 + 0] * Rk[0 * 3 + j] + cur_cw[i * 4 + 1] * Rk[1 * 3 + j] + cur_cw[i * 4 + 2] * Rk[2 * 3 + j];
    kf_keep[k] = rotation_angle(Rd) > 0.5 ? 0 : 1;
  }
}
} // namespace dsm
using dsm::trim_keyframes;
extern "C" {

int dsm_generate_spherical_points(int n_kf, const int *kf_ids, const double *kf_pose_wc, const double *cur_cw, double lidar_range,
                                  int n_pts, const int *pt_kf_id, const double *pt_xyz, int *kf_keep, int *n_out, int *sel_idx,
                                  double *pts_spherical) {
  if (n_kf < 0 || n_pts < 0 || !cur_cw || !n_out || !(lidar_range > 0) || (n_kf && (!kf_ids || !kf_pose_wc || !kf_keep)) ||
      (n_pts && (!pt_kf_id || !pt_xyz || !sel_idx || !pts_spherical)))
    return DSM_ERR_INVALID;
  // :33-41 keyframes whose orientation differs from the current one by more than 0.5 rad are trimmed
  std::vector<std::pair<int, int>> keep_ids; // (id, kept)
  trim_keyframes(n_kf, kf_pose_wc, cur_cw, kf_keep);
  for (int k = 0; k < n_kf; k++) keep_ids.push_back(std::make_pair(kf_ids[k], kf_keep[k]));
  std::sort(keep_ids.begin(), keep_ids.end());
  auto kept = [&](int id) {
    auto it = std::lower_bound(keep_ids.begin(), keep_ids.end(), std::make_pair(id, 0));
    for (; it != keep_ids.end() && it->first == id; ++it)
      if (it->second) return true;
    return false; // unknown keyframe: `find == end` (:55)
  };
  // :44-50
  const double steps[3] = {1.0 / 1.0, 1.0 / 0.5, 1.0 / 1.0}; // RES_X, RES_Y, RES_Z (:23-25)
  const long long vs0 = (long long)std::floor(2 * lidar_range * steps[0]) + 1, vs1 = (long long)std::floor(2 * lidar_range * steps[1]) + 1;
  struct Cell {
    int idx;
    double p[3];
  };
  std::vector<std::pair<long long, Cell>> cells;
  cells.reserve(n_pts);
  for (int i = 0; i < n_pts; i++) { // :52-77
    if (!kept(pt_kf_id[i])) continue;
    const double *g = pt_xyz + 3 * (size_t)i;
    double p[3];
    for (int r = 0; r < 3; r++) p[r] = ((cur_cw[r * 4 + 0] * g[0] + cur_cw[r * 4 + 1] * g[1]) + cur_cw[r * 4 + 2] * g[2]) + cur_cw[r * 4 + 3] * 1.0;
    if (std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) >= lidar_range) continue;
    const long long xi = (long long)std::floor((p[0] + lidar_range) * steps[0]), yi = (long long)std::floor((p[1] + lidar_range) * steps[1]),
                    zi = (long long)std::floor((p[2] + lidar_range) * steps[2]);
    Cell c;
    c.idx = i, c.p[0] = p[0], c.p[1] = p[1], c.p[2] = p[2];
    cells.push_back(std::make_pair(xi + yi * vs0 + zi * vs0 * vs1, c));
  }
  // "store the highest points" (:73-76): per voxel the point with the smallest y; the first one wins ties.  The reference
  // emits its unordered_map in implementation-defined order; here: ascending voxel index (ScanContext::generate does not
  // depend on the order beyond the rounding of its PCA sums).
  std::stable_sort(cells.begin(), cells.end(), [](const std::pair<long long, Cell> &a, const std::pair<long long, Cell> &b) { return a.first < b.first; });
  int n = 0;
  for (size_t a = 0; a < cells.size();) {
    size_t b = a, win = a;
    for (; b < cells.size() && cells[b].first == cells[a].first; b++)
      if (cells[b].second.p[1] < cells[win].second.p[1]) win = b; // -stored.y < -p.y
    sel_idx[n] = cells[win].second.idx;
    for (int r = 0; r < 3; r++) pts_spherical[3 * (size_t)n + r] = cells[win].second.p[r];
    n++;
    a = b;
  }
  *n_out = n;
  return DSM_OK;
}

// ---------------------------------------------------------------------------------------------
// makeCoarseDepthL0, TrackerAndScaler.cpp:143-315 (flat-array form)
// ---------------------------------------------------------------------------------------------
int dsm_make_coarse_depth_l0(int w0, int h0, int nl, int npts, const float *pu, const float *pv, const float *pidepth,
                             const float *pweight, const float *const *ref_dIp, int *n_out, float *const *pc_u,
                             float *const *pc_v, float *const *pc_idepth, float *const *pc_color) {
  if (nl < 1 || nl > DSM_MAX_LEVELS || npts < 0 || !ref_dIp || !n_out || !pc_u || !pc_v || !pc_idepth || !pc_color)
    return DSM_ERR_INVALID;
  std::vector<std::vector<float>> idepth(nl), wsum(nl), bak(nl);
  int w[DSM_MAX_LEVELS], h[DSM_MAX_LEVELS];
  for (int l = 0; l < nl; l++) {
    w[l] = w0 >> l;
    h[l] = h0 >> l;
    idepth[l].assign((size_t)w[l] * h[l], 0.f);
    wsum[l].assign((size_t)w[l] * h[l], 0.f);
    bak[l].assign((size_t)w[l] * h[l], 0.f);
  }
  for (int k = 0; k < npts; k++) { // :149-164
    const int u = pu[k] + 0.5f;
    const int v = pv[k] + 0.5f;
    if (u < 0 || v < 0 || u >= w[0] || v >= h[0]) return DSM_ERR_INVALID; // the reference would write out of bounds
    idepth[0][u + w[0] * v] += pidepth[k] * pweight[k];
    wsum[0][u + w[0] * v] += pweight[k];
  }
  for (int lvl = 1; lvl < nl; lvl++) { // :166-187 (2x2 sums)
    const int wl = w[lvl], hl = h[lvl], wlm1 = w[lvl - 1];
    const float *idm = idepth[lvl - 1].data(), *wsm = wsum[lvl - 1].data();
    for (int y = 0; y < hl; y++)
      for (int x = 0; x < wl; x++) {
        const int b = 2 * x + 2 * y * wlm1;
        idepth[lvl][x + y * wl] = idm[b] + idm[b + 1] + idm[b + wlm1] + idm[b + wlm1 + 1];
        wsum[lvl][x + y * wl] = wsm[b] + wsm[b + 1] + wsm[b + wlm1] + wsm[b + wlm1 + 1];
      }
  }
  for (int lvl = 0; lvl < nl; lvl++) { // dilation :190-275
    const int wl = w[lvl], wh = w[lvl] * h[lvl] - w[lvl];
    float *ws = wsum[lvl].data(), *bk = bak[lvl].data(), *idl = idepth[lvl].data();
    memcpy(bk, ws, sizeof(float) * (size_t)w[lvl] * h[lvl]);
    const int diag[4] = {1 + wl, -1 - wl, wl - 1, -wl + 1}, axis[4] = {1, -1, wl, -wl};
    const int *off = lvl < 2 ? diag : axis;
    for (int i = wl; i < wh; i++)
      if (bk[i] <= 0) {
        float sum = 0, num = 0, numn = 0;
        for (int k = 0; k < 4; k++)
          if (bk[i + off[k]] > 0) {
            sum += idl[i + off[k]];
            num += bk[i + off[k]];
            numn++;
          }
        if (numn > 0) {
          idl[i] = sum / numn;
          ws[i] = num / numn;
        }
      }
  }
  for (int lvl = 0; lvl < nl; lvl++) { // :278-314
    float *ws = wsum[lvl].data(), *idl = idepth[lvl].data();
    const float *ref = ref_dIp[lvl];
    const int wl = w[lvl], hl = h[lvl];
    int n = 0;
    for (int y = 2; y < hl - 2; y++)
      for (int x = 2; x < wl - 2; x++) {
        const int i = x + y * wl;
        if (ws[i] > 0) {
          idl[i] /= ws[i];
          pc_u[lvl][n] = x;
          pc_v[lvl][n] = y;
          pc_idepth[lvl][n] = idl[i];
          pc_color[lvl][n] = ref[3 * i];
          if (!std::isfinite(pc_color[lvl][n]) || !(idl[i] > 0)) {
            idl[i] = -1;
            continue;
          }
          n++;
        } else
          idl[i] = -1;
        ws[i] = 1;
      }
    n_out[lvl] = n;
  }
  return DSM_OK;
}

} // extern "C"
