// loopdet_internal.hpp -- host math shared by host_capi.cpp (the host forms) and loopdet_kernels.hip (the device forms) of
// generate_spherical_points / ScanContext::generate
#pragma once

namespace dsm {
// symmetric 3x3 eigen-decomposition, eigenvalues ascending, eigenvectors in the columns of V, each oriented so that its
// largest-magnitude component is positive (ScanContext.cpp:41-47; Eigen leaves the sign undefined)
void eig3_sym(const double A[9], double evals[3], double V[9]);
// generate_spherical_points.h:33-41: kf_keep[k] = 0 where keyframe k is rotated by more than 0.5 rad against the current one
void trim_keyframes(int n_kf, const double *kf_pose_wc, const double *cur_cw, int *kf_keep);
} // namespace dsm
