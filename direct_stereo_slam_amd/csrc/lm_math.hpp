// lm_math.hpp -- double precision device math of the LM drivers: Sophus SE3 exp / product,
// Eigen quaternion->rotation, DSO AffLight::fromToVecExposure (the pivoted LDLT solve is wave-parallel
// and lives in tracker_kernels.hip).
// These are the un-vendored third-party pieces the reference calls at
// TrackerAndScaler.cpp:509-534 (ldlt().solve), :551 (SE3::exp, operator*), :715/:1023
// (rotationMatrix), :647-649/:717-720 (fromToVecExposure); restated from their published
// algorithms.  Compiled with -ffp-contract=off so that the float results handed to the
// per-point kernels do not depend on FMA contraction.
#pragma once
#include <hip/hip_runtime.h>

namespace dsm {

__device__ inline void quat_normalize(double q[4]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n;
  q[1] /= n;
  q[2] /= n;
  q[3] /= n;
}

// Eigen::Quaterniond::toRotationMatrix
__device__ inline void quat_to_rot(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

__device__ inline void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

// Eigen Quaternion::_transformVector
__device__ inline void quat_rotate(const double q[4], const double v[3], double o[3]) {
  double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
  uv0 += uv0;
  uv1 += uv1;
  uv2 += uv2;
  const double c0 = q[1] * uv2 - q[2] * uv1, c1 = q[2] * uv0 - q[0] * uv2, c2 = q[0] * uv1 - q[1] * uv0;
  o[0] = v[0] + q[3] * uv0 + c0;
  o[1] = v[1] + q[3] * uv1 + c1;
  o[2] = v[2] + q[3] * uv2 + c2;
}

// Sophus SE3 product a*b with pose = {qx,qy,qz,qw,tx,ty,tz}
__device__ inline void se3_mul(const double a[7], const double b[7], double out[7]) {
  double r[3], q[4];
  quat_rotate(a, b + 4, r);
  quat_mul(a, b, q);
  quat_normalize(q);
  out[0] = q[0];
  out[1] = q[1];
  out[2] = q[2];
  out[3] = q[3];
  out[4] = a[4] + r[0];
  out[5] = a[5] + r[1];
  out[6] = a[6] + r[2];
}

// value of `v` in lane `l` (wave-uniform l): v_readlane, no LDS round trip
__device__ __forceinline__ double lane_value_d(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Sophus SE3::exp, tangent = [upsilon ; omega], executed by a whole wave on wave-uniform input: lanes 0
// and 1 evaluate sincos(theta/2) and sincos(theta) side by side, everything else is computed redundantly
// by all lanes.
__device__ inline void se3_exp_wave(const double xi[6], double pose[7], int lane) {
  const double *ups = xi, *om = xi + 3;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double theta = sqrt(theta_sq);
  const double half = 0.5 * theta;
  double imag, real;
  const double eps = 1e-10;
  double s_half = 0, c_half = 1, s_th = 0, c_th = 1;
  if (!(theta < eps)) {
    double sv, cv;
    sincos((lane & 1) ? theta : half, &sv, &cv);
    s_half = lane_value_d(sv, 0), c_half = lane_value_d(cv, 0);
    s_th = lane_value_d(sv, 1), c_th = lane_value_d(cv, 1);
  }
  if (theta < eps) {
    const double t2 = theta * theta, t4 = t2 * t2;
    imag = 0.5 - (1.0 / 48.0) * t2 + (1.0 / 3840.0) * t4;
    real = 1.0 - 0.5 * t2 + (1.0 / 384.0) * t4;
  } else {
    imag = s_half / theta;
    real = c_half;
  }
  double q[4] = {imag * om[0], imag * om[1], imag * om[2], real};
  quat_normalize(q);
  double V[9];
  if (theta < eps) {
    quat_to_rot(q, V);
  } else {
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        O2[i * 3 + j] = O[i * 3 + 0] * O[0 * 3 + j] + O[i * 3 + 1] * O[1 * 3 + j] + O[i * 3 + 2] * O[2 * 3 + j];
    const double ca = (1.0 - c_th) / theta_sq;
    const double cb = (theta - s_th) / (theta_sq * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + ca * O[i] + cb * O2[i];
  }
  pose[0] = q[0];
  pose[1] = q[1];
  pose[2] = q[2];
  pose[3] = q[3];
#pragma unroll
  for (int i = 0; i < 3; i++) pose[4 + i] = V[i * 3 + 0] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

// DSO AffLight::fromToVecExposure
__device__ inline void aff_from_to(float expF, float expT, double g2F_a, double g2F_b, double g2T_a,
                                   double g2T_b, double out[2]) {
  if (expF == 0 || expT == 0) expT = expF = 1;
  const double a = exp(g2T_a - g2F_a) * expT / expF;
  out[0] = a;
  out[1] = g2T_b - a * g2F_b;
}

// 3x3 float product, ((a0*b0 + a1*b1) + a2*b2)
__device__ inline void mat3f_mul(const float *a, const float *b, float *o) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      o[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}

} // namespace dsm
