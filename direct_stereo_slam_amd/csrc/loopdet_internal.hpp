// loopdet_internal.hpp -- host math shared by host_capi.cpp (the host forms) and loopdet_kernels.hip (the device forms) of
// generate_spherical_points / ScanContext::generate
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace dsm {
// Summation order of align_points_PCA's moments, shared by the host form (host_capi.cpp) and the device form (loopdet_kernels.hip):
//   * the MEAN is the reference's own loop, three sums in point order (ScanContext.cpp:22-29) -- kept to the letter;
//   * the COVARIANCE `pts_mat.transpose() * pts_mat` (:40) is an Eigen product whose accumulation order the reference does not define
//     (blocked GEMM): here kCovLanes interleaved partial sums -- partial l adds the products of points l, l + kCovLanes, ... in that
//     order -- which are then added in ascending l.  (Round 5 summed the covariance in point order too, on ONE lane per moment: 0.86 of
//     the 0.97 ms a single keyframe's loop chain took on the device.)
constexpr int kCovLanes = 256;

// symmetric 3x3 eigen-decomposition, eigenvalues ascending, eigenvectors in the columns of V, each oriented so that its
// largest-magnitude component is positive (ScanContext.cpp:41-47; Eigen leaves the sign undefined).  Host AND device: the
// device form of the loop descriptors (loopdet_kernels.hip) runs the same operations in the same order (-ffp-contract=off).
// symmetric 3x3 eigen-decomposition (cyclic Jacobi, double), eigenvalues ascending as
// Eigen::SelfAdjointEigenSolver returns them (:43-47); columns of V are the eigenvectors
__host__ __device__ inline void eig3_sym(const double A_in[9], double evals[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; i++) A[i] = A_in[i];
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { // A <- A J
          const double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) { // A <- J^T A
          const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) { // V <- V J
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  const double d[3] = {A[0], A[4], A[8]};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (d[idx[j]] > d[idx[j + 1]]) {
        const int t = idx[j];
        idx[j] = idx[j + 1];
        idx[j + 1] = t;
      }
  double Vs[9];
  for (int c = 0; c < 3; c++) {
    evals[c] = d[idx[c]];
    int big = 0;
    for (int r = 1; r < 3; r++)
      if (fabs(V[r * 3 + idx[c]]) > fabs(V[big * 3 + idx[c]])) big = r;
    const double sgn = V[big * 3 + idx[c]] < 0 ? -1.0 : 1.0; // orientation convention, see header
    for (int r = 0; r < 3; r++) Vs[r * 3 + c] = sgn * V[r * 3 + idx[c]];
  }
  for (int i = 0; i < 9; i++) V[i] = Vs[i];
}

// generate_spherical_points.h:33-41: kf_keep[k] = 0 where keyframe k is rotated by more than 0.5 rad against the current one
void trim_keyframes(int n_kf, const double *kf_pose_wc, const double *cur_cw, int *kf_keep);
} // namespace dsm
