/*
 * dsm_oracle.c -- CPU restatement ("oracle") of the hot path.  TEST INFRASTRUCTURE ONLY.
 * See dsm_oracle.h for the usage rule and the "PARITY UNPINNED" statement.
 *
 * Build (parity): gcc -O2 -ffp-contract=off -fno-fast-math   (oracle/Makefile)
 * Build (timing): gcc -O3 -march=native                       (mirrors CMakeLists.txt:4-6)
 *
 * All citations are relative to the reference tree (/root/reference).
 * "UPSTREAM" marks arithmetic that lives in un-vendored dependencies (DSO, Sophus, Eigen,
 * FLANN); it is restated from the published algorithm and the reference's call sites.
 */
#include "dsm_oracle_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                         */
/* ------------------------------------------------------------------------------------ */
static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}

void orc_params_default(orc_params *p) {
  /* upstream DSO settings.cpp defaults; affine modes as set by mode=1 (src/main.cpp:117-121) */
  p->huber_th = 9.0f;
  p->coarse_cutoff_th = 20.0f;
  p->scale_xi_rot = 1.0f;
  p->scale_xi_trans = 0.5f;
  p->scale_a = 10.0f;
  p->scale_b = 1000.0f;
  p->affine_opt_mode_a = 0.0f;
  p->affine_opt_mode_b = 0.0f;
  p->lambda_extrapolation_limit = 0.001f; /* TrackerAndScaler.cpp:464,863 */
  const int it[ORC_MAX_LEVELS] = {10, 20, 50, 50, 50, 50}; /* :463,:862 ([5] is an extension) */
  memcpy(p->max_iterations, it, sizeof it);
  p->fixed_schedule = 0;
}

/* ------------------------------------------------------------------------------------ */
/* Sophus / Eigen geometry restatements (UPSTREAM), all double                           */
/* pose = {qx,qy,qz,qw, tx,ty,tz}                                                        */
/* ------------------------------------------------------------------------------------ */
static void quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n;
  q[1] /= n;
  q[2] /= n;
  q[3] /= n;
}

/* Eigen Quaternion::toRotationMatrix (used by Sophus SO3::matrix / SE3::rotationMatrix,
 * call sites TrackerAndScaler.cpp:715,1023) */
void orc_quat_to_rot(const double q[4], double R[9]) {
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

/* Eigen quaternion product a*b */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

/* Eigen Quaternion::_transformVector: v + w*uv + q.vec x uv with uv = 2 (q.vec x v) */
static void quat_rotate(const double q[4], const double v[3], double o[3]) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2],
                       q[0] * uv[1] - q[1] * uv[0]};
  o[0] = v[0] + q[3] * uv[0] + c[0];
  o[1] = v[1] + q[3] * uv[1] + c[1];
  o[2] = v[2] + q[3] * uv[2] + c[2];
}

/* Sophus SE3Group::operator* : t = t_a + R_a t_b ; q = normalize(q_a q_b) (call site :551) */
void orc_se3_mul(const double a[7], const double b[7], double out[7]) {
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

/* Sophus SE3Group::exp (tangent = [upsilon(3) ; omega(3)]), call site TrackerAndScaler.cpp:551 */
void orc_se3_exp(const double xi[6], double pose[7]) {
  const double *ups = xi, *om = xi + 3;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double theta = sqrt(theta_sq);
  const double half = 0.5 * theta;
  double imag, real;
  const double eps = 1e-10; /* SophusConstants<double>::epsilon() */
  if (theta < eps) {
    const double t2 = theta * theta, t4 = t2 * t2;
    imag = 0.5 - (1.0 / 48.0) * t2 + (1.0 / 3840.0) * t4;
    real = 1.0 - 0.5 * t2 + (1.0 / 384.0) * t4;
  } else {
    imag = sin(half) / theta;
    real = cos(half);
  }
  double q[4] = {imag * om[0], imag * om[1], imag * om[2], real};
  quat_normalize(q);
  /* V = I + (1-cos)/theta^2 * Omega + (theta - sin)/theta^3 * Omega^2 ; V = R if theta < eps */
  double V[9];
  if (theta < eps) {
    orc_quat_to_rot(q, V);
  } else {
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        O2[i * 3 + j] = O[i * 3 + 0] * O[0 * 3 + j] + O[i * 3 + 1] * O[1 * 3 + j] + O[i * 3 + 2] * O[2 * 3 + j];
    const double ca = (1.0 - cos(theta)) / theta_sq;
    const double cb = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + ca * O[i] + cb * O2[i];
  }
  pose[0] = q[0];
  pose[1] = q[1];
  pose[2] = q[2];
  pose[3] = q[3];
  for (int i = 0; i < 3; i++) pose[4 + i] = V[i * 3 + 0] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

/* SE3(Matrix4d) as used for tfm_f1_f0_ (TrackerAndScaler.cpp:82-86): Eigen rotation-matrix ->
 * quaternion, normalised by the Sophus SO3 constructor. */
void orc_se3_from_matrix(const double T[16], double pose[7]) {
  const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6];
  const double m20 = T[8], m21 = T[9], m22 = T[10];
  double q[4];
  double t = m00 + m11 + m22;
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m21 - m12) * t;
    q[1] = (m02 - m20) * t;
    q[2] = (m10 - m01) * t;
  } else {
    const double M[3][3] = {{m00, m01, m02}, {m10, m11, m12}, {m20, m21, m22}};
    int i = 0;
    if (m11 > m00) i = 1;
    if (m22 > M[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(M[i][i] - M[j][j] - M[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M[k][j] - M[j][k]) * t;
    q[j] = (M[j][i] + M[i][j]) * t;
    q[k] = (M[k][i] + M[i][k]) * t;
  }
  quat_normalize(q);
  pose[0] = q[0];
  pose[1] = q[1];
  pose[2] = q[2];
  pose[3] = q[3];
  pose[4] = T[3];
  pose[5] = T[7];
  pose[6] = T[11];
}

/* Eigen LDLT<Lower> (pivoting on the largest |diagonal|) + solve, UPSTREAM, call sites
 * TrackerAndScaler.cpp:509,513,518,529.  A is row-major n x n, only its lower triangle is used. */
void orc_ldlt_solve(int n, const double *Ain, const double *rhs, double *x) {
  double A[64], temp[8], d[8];
  int tr[8];
  if (n > 8) abort();
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[i * 8 + j] = Ain[i * n + j];
#define A_(i, j) A[(i) * 8 + (j)]
  int all_zero = 0;
  for (int k = 0; k < n; k++) {
    int big = k;
    double bigv = fabs(A_(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(A_(i, i)) > bigv) {
        bigv = fabs(A_(i, i));
        big = i;
      }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; j++) {
        double s = A_(k, j);
        A_(k, j) = A_(big, j);
        A_(big, j) = s;
      }
      for (int i = big + 1; i < n; i++) {
        double s = A_(i, k);
        A_(i, k) = A_(i, big);
        A_(i, big) = s;
      }
      {
        double s = A_(k, k);
        A_(k, k) = A_(big, big);
        A_(big, big) = s;
      }
      for (int i = k + 1; i < big; i++) {
        double s = A_(i, k);
        A_(i, k) = A_(big, i);
        A_(big, i) = s;
      }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = A_(j, j) * A_(k, j);
      double dot = 0;
      for (int j = 0; j < k; j++) dot += A_(k, j) * temp[j];
      A_(k, k) -= dot;
      for (int i = 0; i < rs; i++) {
        double s = 0;
        for (int j = 0; j < k; j++) s += A_(k + 1 + i, j) * temp[j];
        A_(k + 1 + i, k) -= s;
      }
    }
    const double akk = A_(k, k);
    const int valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      all_zero = 1;
      break;
    }
    if (rs > 0 && valid)
      for (int i = 0; i < rs; i++) A_(k + 1 + i, k) /= akk;
  }
  double y[8];
  for (int i = 0; i < n; i++) y[i] = rhs[i];
  if (!all_zero) {
    for (int k = 0; k < n; k++) { /* dst = P rhs */
      double s = y[k];
      y[k] = y[tr[k]];
      y[tr[k]] = s;
    }
    for (int i = 0; i < n; i++) /* L^-1 */
      for (int j = 0; j < i; j++) y[i] -= A_(i, j) * y[j];
    for (int i = 0; i < n; i++) d[i] = A_(i, i);
    const double tol = 1.0 / 1.7976931348623157e308; /* 1 / NumTraits<double>::highest() */
    for (int i = 0; i < n; i++) {
      if (fabs(d[i]) > tol)
        y[i] /= d[i];
      else
        y[i] = 0;
    }
    for (int i = n - 1; i >= 0; i--) /* L^-T */
      for (int j = i + 1; j < n; j++) y[i] -= A_(j, i) * y[j];
    for (int k = n - 1; k >= 0; k--) { /* P^T */
      double s = y[k];
      y[k] = y[tr[k]];
      y[tr[k]] = s;
    }
  } else {
    for (int i = 0; i < n; i++) y[i] = 0;
  }
  for (int i = 0; i < n; i++) x[i] = y[i];
#undef A_
}

/* ------------------------------------------------------------------------------------ */
/* float helpers                                                                         */
/* ------------------------------------------------------------------------------------ */
/* Eigen Matrix3f::inverse() (cofactor form), call site TrackerAndScaler.cpp:139 */
static float cof3(const float *m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void mat3f_inverse(const float *m, float *r) {
  const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  const float invdet = 1.0f / det;
  r[0] = c0 * invdet;
  r[1] = c1 * invdet;
  r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet;
  r[4] = cof3(m, 1, 1) * invdet;
  r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet;
  r[7] = cof3(m, 1, 2) * invdet;
  r[8] = cof3(m, 2, 2) * invdet;
}
static void mat3f_mul(const float *a, const float *b, float *o) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      o[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}

/* AffLight::fromToVecExposure (UPSTREAM DSO util/NumType.h), call sites :647-649,:717-720 */
void orc_aff_from_to(float expF, float expT, double g2F_a, double g2F_b, double g2T_a, double g2T_b, double out[2]) {
  if (expF == 0 || expT == 0) expT = expF = 1;
  const double a = exp(g2T_a - g2F_a) * expT / expF;
  const double b = g2T_b - a * g2F_b;
  out[0] = a;
  out[1] = b;
}

/* getInterpolatedElement33 (UPSTREAM DSO util/globalFuncs.h), call sites :790,:1106 */
static inline void interp33(const float *mat, float x, float y, int width, float out[3]) {
  const int ix = (int)x;
  const int iy = (int)y;
  const float dx = x - ix;
  const float dy = y - iy;
  const float dxdy = dx * dy;
  const float *bp = mat + 3 * (ix + iy * width);
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    out[c] = ((w11 * bp[3 * (1 + width) + c] + w01 * bp[3 * width + c]) + w10 * bp[3 + c]) + w00 * bp[c];
}

/* ------------------------------------------------------------------------------------ */
/* tracker state                                                                         */
/* ------------------------------------------------------------------------------------ */
orc_tracker *orc_tracker_create(int ww, int hh, int nlevels, const double T[16], const float K1[4],
                                const orc_params *p) {
  orc_tracker *t = (orc_tracker *)xcalloc(1, sizeof *t);
  t->p = *p;
  t->nlevels = nlevels;
  for (int l = 0; l < nlevels; l++) { /* TrackerAndScaler.cpp:52-64 */
    const int wl = ww >> l, hl = hh >> l;
    t->w[l] = wl;
    t->h[l] = hl;
    t->idepth[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->wsum[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->wsum_bak[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->pc_u[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->pc_v[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->pc_id[l] = (float *)xcalloc((size_t)wl * hl, 4);
    t->pc_c[l] = (float *)xcalloc((size_t)wl * hl, 4);
  }
  for (int i = 0; i < 8; i++) { /* :67-74, :101-108 */
    t->pb[i] = (float *)xcalloc((size_t)ww * hh + 4, 4);
    t->sb[i] = (float *)xcalloc((size_t)ww * hh + 4, 4);
  }
  t->ref_id = -1;
  orc_se3_from_matrix(T, t->T10); /* :82-86 */
  t->fx1[0] = K1[0];              /* :89-98 */
  t->fy1[0] = K1[1];
  t->cx1[0] = K1[2];
  t->cy1[0] = K1[3];
  for (int l = 1; l < nlevels; l++) {
    t->fx1[l] = t->fx1[l - 1] * 0.5;
    t->fy1[l] = t->fy1[l - 1] * 0.5;
    t->cx1[l] = (t->cx1[0] + 0.5) / ((int)1 << l) - 0.5;
    t->cy1[l] = (t->cy1[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  return t;
}

void orc_tracker_destroy(orc_tracker *t) {
  if (!t) return;
  for (int l = 0; l < t->nlevels; l++) {
    free(t->idepth[l]);
    free(t->wsum[l]);
    free(t->wsum_bak[l]);
    free(t->pc_u[l]);
    free(t->pc_v[l]);
    free(t->pc_id[l]);
    free(t->pc_c[l]);
  }
  for (int i = 0; i < 8; i++) {
    free(t->pb[i]);
    free(t->sb[i]);
  }
  free(t);
}

/* TrackerAndScaler::makeK, :117-141 */
void orc_tracker_make_k(orc_tracker *t, float fx, float fy, float cx, float cy) {
  t->fx[0] = fx;
  t->fy[0] = fy;
  t->cx[0] = cx;
  t->cy[0] = cy;
  for (int l = 1; l < t->nlevels; l++) {
    t->fx[l] = t->fx[l - 1] * 0.5;
    t->fy[l] = t->fy[l - 1] * 0.5;
    t->cx[l] = (t->cx[0] + 0.5) / ((int)1 << l) - 0.5;
    t->cy[l] = (t->cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  for (int l = 0; l < t->nlevels; l++) {
    const float K[9] = {t->fx[l], 0.0f, t->cx[l], 0.0f, t->fy[l], t->cy[l], 0.0f, 0.0f, 1.0f};
    mat3f_inverse(K, t->Ki[l]);
  }
}

/* result of setCoarseTrackingRef, :317-327 (the lists come from makeCoarseDepthL0 or the caller) */
void orc_tracker_set_ref(orc_tracker *t, int ref_id, double ref_a, double ref_b, float ref_exposure,
                         const int *n, const float *const *u, const float *const *v,
                         const float *const *id, const float *const *c) {
  for (int l = 0; l < t->nlevels; l++) {
    t->pc_n[l] = n[l];
    memcpy(t->pc_u[l], u[l], sizeof(float) * n[l]);
    memcpy(t->pc_v[l], v[l], sizeof(float) * n[l]);
    memcpy(t->pc_id[l], id[l], sizeof(float) * n[l]);
    memcpy(t->pc_c[l], c[l], sizeof(float) * n[l]);
  }
  t->ref_id = ref_id;
  t->ref_a = ref_a;
  t->ref_b = ref_b;
  t->ref_exposure = ref_exposure;
}

/* scaleCoarseDepthL0, :329-336 */
void orc_tracker_scale_depth(orc_tracker *t, float scale) {
  for (int l = 0; l < t->nlevels; l++)
    for (int p = 0; p < t->pc_n[l]; p++) t->pc_id[l][p] /= scale;
}

int orc_tracker_get_template(orc_tracker *t, int l, int *n, float *u, float *v, float *id, float *c) {
  *n = t->pc_n[l];
  if (u) memcpy(u, t->pc_u[l], 4 * (size_t)*n);
  if (v) memcpy(v, t->pc_v[l], 4 * (size_t)*n);
  if (id) memcpy(id, t->pc_id[l], 4 * (size_t)*n);
  if (c) memcpy(c, t->pc_c[l], 4 * (size_t)*n);
  return 0;
}

void orc_tracker_set_frame(orc_tracker *t, int slot, const float *const *dIp, float ab_exposure) {
  for (int l = 0; l < t->nlevels; l++) t->dIp[slot][l] = dIp[l];
  t->exposure[slot] = ab_exposure;
}

void orc_tracker_use_sse(orc_tracker *t, int on) { t->use_sse = on; }
int orc_pose_warped_n(orc_tracker *t) { return t->pb_n; }
double orc_last_energy_f64(orc_tracker *t) { return t->last_E_f64; }
int orc_scale_warped_n(orc_tracker *t) { return t->sb_n; }
void orc_get_eval_counts(orc_tracker *t, int64_t r[ORC_MAX_LEVELS], int64_t g[ORC_MAX_LEVELS]) {
  memcpy(r, t->res_evals, sizeof t->res_evals);
  memcpy(g, t->gs_evals, sizeof t->gs_evals);
}

/* ------------------------------------------------------------------------------------ */
/* calcResPose, TrackerAndScaler.cpp:699-852                                             */
/* ------------------------------------------------------------------------------------ */
void orc_calc_res_pose(orc_tracker *t, int lvl, const double pose[7], const double aff[2],
                       float cutoffTH, double rs[6]) {
  float E = 0;
  int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
  const int wl = t->w[lvl], hl = t->h[lvl];
  const float *dINewl = t->dIp[0][lvl];
  const float fxl = t->fx[lvl], fyl = t->fy[lvl], cxl = t->cx[lvl], cyl = t->cy[lvl];
  const float *Ki = t->Ki[lvl];

  double Rd[9];
  orc_quat_to_rot(pose, Rd);
  float Rf[9], RKi[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
  mat3f_mul(Rf, Ki, RKi);                                                     /* :715 */
  const float tt[3] = {(float)pose[4], (float)pose[5], (float)pose[6]};       /* :716 */
  double affd[2];
  orc_aff_from_to(t->ref_exposure, t->exposure[0], t->ref_a, t->ref_b, aff[0], aff[1], affd); /* :717-720 */
  const float affLL0 = (float)affd[0], affLL1 = (float)affd[1];

  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  const float huberTH = t->p.huber_th;
  const float maxEnergy = 2 * huberTH * cutoffTH - huberTH * huberTH; /* :726-728 */

  const int nl = t->pc_n[lvl];
  const float *lpc_u = t->pc_u[lvl], *lpc_v = t->pc_v[lvl], *lpc_idepth = t->pc_id[lvl],
              *lpc_color = t->pc_c[lvl];
  float **B = t->pb;
  t->res_evals[lvl]++;
  double E64 = 0;

  for (int i = 0; i < nl; i++) {
    const float id = lpc_idepth[i], x = lpc_u[i], y = lpc_v[i];
    float pt[3];
    for (int r = 0; r < 3; r++) pt[r] = ((RKi[r * 3] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2]) + tt[r] * id; /* :747 */
    const float u = pt[0] / pt[2];
    const float v = pt[1] / pt[2];
    const float Ku = fxl * u + cxl;
    const float Kv = fyl * v + cyl;
    const float new_idepth = id / pt[2];

    if (lvl == 0 && i % 32 == 0) { /* :754-784 */
      float ptT[3], ptT2[3], pt3[3];
      for (int r = 0; r < 3; r++) {
        const float kx = (Ki[r * 3] * x + Ki[r * 3 + 1] * y) + Ki[r * 3 + 2];
        const float rx = (RKi[r * 3] * x + RKi[r * 3 + 1] * y) + RKi[r * 3 + 2];
        ptT[r] = kx + tt[r] * id;
        ptT2[r] = kx - tt[r] * id;
        pt3[r] = rx - tt[r] * id;
      }
      const float KuT = fxl * (ptT[0] / ptT[2]) + cxl, KvT = fyl * (ptT[1] / ptT[2]) + cyl;
      const float KuT2 = fxl * (ptT2[0] / ptT2[2]) + cxl, KvT2 = fyl * (ptT2[1] / ptT2[2]) + cyl;
      const float Ku3 = fxl * (pt3[0] / pt3[2]) + cxl, Kv3 = fyl * (pt3[1] / pt3[2]) + cyl;
      sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      sumSquaredShiftNum += 2;
    }

    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue; /* :786 */

    const float refColor = lpc_color[i];
    float hit[3];
    interp33(dINewl, Ku, Kv, wl, hit); /* :790 */
    if (!isfinite(hit[0])) continue;
    const float residual = hit[0] - (float)(affLL0 * refColor + affLL1);              /* :793 */
    const float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);       /* :794-795 */

    if (fabsf(residual) > cutoffTH) { /* :797-802 */
      E += maxEnergy;
      E64 += maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw); /* :809 */
      E64 += hw * residual * residual * (2 - hw);
      numTermsInE++;
      B[0][numTermsInWarped] = new_idepth; /* :812-819 */
      B[1][numTermsInWarped] = u;
      B[2][numTermsInWarped] = v;
      B[3][numTermsInWarped] = hit[1];
      B[4][numTermsInWarped] = hit[2];
      B[5][numTermsInWarped] = residual;
      B[6][numTermsInWarped] = hw;
      B[7][numTermsInWarped] = lpc_color[i];
      numTermsInWarped++;
    }
  }
  while (numTermsInWarped % 4 != 0) { /* :824-834 */
    for (int k = 0; k < 8; k++) B[k][numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  t->pb_n = numTermsInWarped;
  t->last_E_f64 = E64;

  rs[0] = E; /* :843-851 */
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
}

/* ------------------------------------------------------------------------------------ */
/* Accumulator emulation (UPSTREAM DSO Accumulator9; same scheme as ScaleAccumulator.h)  */
/* 4 SSE lanes, shifted into a "1k" then "1m" buffer (quirk Q4).  Note: num_in_1k counts  */
/* packs, so it exceeds 1000 after the first shift and the 1k buffer is forwarded to 1m   */
/* immediately -- reproduced as written (ScaleAccumulator.h:85-105).                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  int nent;
  float d[45][4], d1k[45][4], d1m[45][4];
  float n1, n1k, n1m;
} lane_acc;

static void acc_init(lane_acc *a, int nent) {
  memset(a, 0, sizeof *a);
  a->nent = nent;
}
static void acc_shift(lane_acc *a, int force) {
  if (a->n1 > 1000 || force) {
    for (int i = 0; i < a->nent; i++)
      for (int k = 0; k < 4; k++) {
        a->d1k[i][k] = a->d[i][k] + a->d1k[i][k];
        a->d[i][k] = 0;
      }
    a->n1k += a->n1;
    a->n1 = 0;
  }
  if (a->n1k > 1000 || force) {
    for (int i = 0; i < a->nent; i++)
      for (int k = 0; k < 4; k++) {
        a->d1m[i][k] = a->d1k[i][k] + a->d1m[i][k];
        a->d1k[i][k] = 0;
      }
    a->n1m += a->n1k;
    a->n1k = 0;
  }
}
static float acc_finish_entry(const lane_acc *a, int idx) {
  return a->d1m[idx][0] + a->d1m[idx][1] + a->d1m[idx][2] + a->d1m[idx][3];
}

/* ------------------------------------------------------------------------------------ */
/* calcGSSSEPose, TrackerAndScaler.cpp:640-697                                           */
/* ------------------------------------------------------------------------------------ */
void orc_calc_gs_pose(orc_tracker *t, int lvl, const double pose[7], const double aff[2],
                      double H_out[64], double b_out[8]) {
  (void)pose;
  if (t->use_sse) {
    orc_calc_gs_pose_sse(t, lvl, aff, H_out, b_out);
    return;
  }
  static lane_acc acc; /* single-threaded test infrastructure */
  acc_init(&acc, 45);
  const float fxl = t->fx[lvl], fyl = t->fy[lvl];
  const float b0 = (float)t->ref_b; /* :646 */
  double affd[2];
  orc_aff_from_to(t->ref_exposure, t->exposure[0], t->ref_a, t->ref_b, aff[0], aff[1], affd);
  const float a = (float)affd[0]; /* :647-649 */
  float **B = t->pb;
  const int n = t->pb_n;
  t->gs_evals[lvl]++;
  for (int i = 0; i < n; i += 4) {
    float J[9][4], w[4];
    for (int k = 0; k < 4; k++) {
      const float dx = B[3][i + k] * fxl; /* :658-662 */
      const float dy = B[4][i + k] * fyl;
      const float u = B[1][i + k], v = B[2][i + k], id = B[0][i + k];
      J[0][k] = id * dx; /* :664-678 */
      J[1][k] = id * dy;
      J[2][k] = 0.0f - id * (u * dx + v * dy);
      J[3][k] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
      J[4][k] = (u * v) * dy + dx * (1.0f + u * u);
      J[5][k] = u * dy - v * dx;
      J[6][k] = a * (b0 - B[7][i + k]);
      J[7][k] = -1.0f;
      J[8][k] = B[5][i + k];
      w[k] = B[6][i + k];
    }
    /* Accumulator9::updateSSE_eighted */
    int idx = 0;
    for (int r = 0; r < 9; r++) {
      float Jw[4];
      for (int k = 0; k < 4; k++) Jw[k] = J[r][k] * w[k];
      for (int c = r; c < 9; c++) {
        for (int k = 0; k < 4; k++) acc.d[idx][k] = acc.d[idx][k] + Jw[k] * J[c][k];
        idx++;
      }
    }
    acc.n1++;
    acc_shift(&acc, 0);
  }
  acc_shift(&acc, 1);
  float Hf[9][9];
  int idx = 0;
  for (int r = 0; r < 9; r++)
    for (int c = r; c < 9; c++) {
      const float d = acc_finish_entry(&acc, idx++);
      Hf[r][c] = Hf[c][r] = d;
    }
  const float invn = 1.0f / n; /* :682-683 */
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H_out[r * 8 + c] = (double)Hf[r][c] * (double)invn;
    b_out[r] = (double)Hf[r][8] * (double)invn;
  }
  /* :685-696 */
  const double s[8] = {t->p.scale_xi_rot,   t->p.scale_xi_rot,   t->p.scale_xi_rot, t->p.scale_xi_trans,
                       t->p.scale_xi_trans, t->p.scale_xi_trans, t->p.scale_a,      t->p.scale_b};
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++) H_out[r * 8 + c] *= s[c];
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++) H_out[r * 8 + c] *= s[r];
  for (int r = 0; r < 8; r++) b_out[r] *= s[r];
}

/* ------------------------------------------------------------------------------------ */
/* trackNewestCoarse, TrackerAndScaler.cpp:451-638                                       */
/* ------------------------------------------------------------------------------------ */
int orc_track(orc_tracker *t, double pose_io[7], double aff_io[2], int coarsestLvl,
              const double *minResForAbort, double *lastResiduals, double flow_out[3]) {
  for (int i = 0; i < ORC_MAX_LEVELS; i++) lastResiduals[i] = NAN; /* :459 */
  flow_out[0] = flow_out[1] = flow_out[2] = 1000;                  /* :460 */
  memset(t->res_evals, 0, sizeof t->res_evals);
  memset(t->gs_evals, 0, sizeof t->gs_evals);
  const int *maxIterations = t->p.max_iterations;
  const int fixed = t->p.fixed_schedule > 0 ? t->p.fixed_schedule : 0; /* benchmark schedule (orc_params), not the reference */
  const float lambdaExtrapolationLimit = t->p.lambda_extrapolation_limit;
  const float cutoff0 = t->p.coarse_cutoff_th;
  double cur[7], aff_cur[2];
  memcpy(cur, pose_io, sizeof cur);
  memcpy(aff_cur, aff_io, sizeof aff_cur);
  int haveRepeated = 0;
  const float modeA = t->p.affine_opt_mode_a, modeB = t->p.affine_opt_mode_b;
  const double sc[8] = {t->p.scale_xi_rot,   t->p.scale_xi_rot,   t->p.scale_xi_rot, t->p.scale_xi_trans,
                        t->p.scale_xi_trans, t->p.scale_xi_trans, t->p.scale_a,      t->p.scale_b};

  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    double H[64], b[8], resOld[6];
    float levelCutoffRepeat = 1;
    orc_calc_res_pose(t, lvl, cur, aff_cur, cutoff0 * levelCutoffRepeat, resOld); /* :475 */
    while (!fixed && resOld[5] > 0.6 && levelCutoffRepeat < 50) {                  /* :477-485 */
      levelCutoffRepeat *= 2;
      orc_calc_res_pose(t, lvl, cur, aff_cur, cutoff0 * levelCutoffRepeat, resOld);
    }
    orc_calc_gs_pose(t, lvl, cur, aff_cur, H, b); /* :487 */
    float lambda = 0.01;

    for (int iteration = 0; iteration < (fixed ? fixed : maxIterations[lvl]); iteration++) {
      double Hl[64], inc[8], nb[8];
      memcpy(Hl, H, sizeof Hl);
      for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda); /* :506-508 */
      for (int i = 0; i < 8; i++) nb[i] = -b[i];
      orc_ldlt_solve(8, Hl, nb, inc); /* :509 */
      if (modeA < 0 && modeB < 0) {   /* :511-515 fix a, b */
        double A6[36], x6[6];
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) A6[r * 6 + c] = Hl[r * 8 + c];
        orc_ldlt_solve(6, A6, nb, x6);
        for (int i = 0; i < 6; i++) inc[i] = x6[i];
        inc[6] = inc[7] = 0;
      }
      if (!(modeA < 0) && modeB < 0) { /* :516-520 fix b */
        double A7[49], x7[7];
        for (int r = 0; r < 7; r++)
          for (int c = 0; c < 7; c++) A7[r * 7 + c] = Hl[r * 8 + c];
        orc_ldlt_solve(7, A7, nb, x7);
        for (int i = 0; i < 7; i++) inc[i] = x7[i];
        inc[7] = 0;
      }
      if (modeA < 0 && !(modeB < 0)) { /* :521-534 fix a */
        double Hs[64], bs[8], A7[49], nbs[7], x7[7];
        memcpy(Hs, Hl, sizeof Hs);
        memcpy(bs, b, sizeof bs);
        for (int r = 0; r < 8; r++) Hs[r * 8 + 6] = Hs[r * 8 + 7];
        for (int c = 0; c < 8; c++) Hs[6 * 8 + c] = Hs[7 * 8 + c];
        bs[6] = bs[7];
        for (int r = 0; r < 7; r++) {
          for (int c = 0; c < 7; c++) A7[r * 7 + c] = Hs[r * 8 + c];
          nbs[r] = -bs[r];
        }
        orc_ldlt_solve(7, A7, nbs, x7);
        for (int i = 0; i < 8; i++) inc[i] = 0;
        for (int i = 0; i < 6; i++) inc[i] = x7[i];
        inc[6] = 0;
        inc[7] = x7[6];
      }
      float extrapFac = 1; /* :536-539 */
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
      for (int i = 0; i < 8; i++) inc[i] *= extrapFac;

      double incScaled[8], sum = 0; /* :541-548 */
      for (int i = 0; i < 8; i++) {
        incScaled[i] = inc[i] * sc[i];
        sum += incScaled[i];
      }
      if (!isfinite(sum))
        for (int i = 0; i < 8; i++) incScaled[i] = 0;

      double ex[7], newp[7], aff_new[2]; /* :550-554 */
      orc_se3_exp(incScaled, ex);
      orc_se3_mul(ex, cur, newp);
      aff_new[0] = aff_cur[0] + incScaled[6];
      aff_new[1] = aff_cur[1] + incScaled[7];

      double resNew[6];
      orc_calc_res_pose(t, lvl, newp, aff_new, cutoff0 * levelCutoffRepeat, resNew); /* :556 */
      const int accept = fixed || (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);  /* :559 */
      if (accept) { /* :576-586 */
        orc_calc_gs_pose(t, lvl, newp, aff_new, H, b);
        memcpy(resOld, resNew, sizeof resOld);
        memcpy(aff_cur, aff_new, sizeof aff_cur);
        memcpy(cur, newp, sizeof cur);
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0;
      for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      nrm = sqrt(nrm);
      if (!fixed && !(nrm > 1e-3)) break; /* :588 */
    }
    lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1])); /* :596 */
    flow_out[0] = resOld[2];                                     /* :597 */
    flow_out[1] = resOld[3];
    flow_out[2] = resOld[4];
    if (!fixed && minResForAbort && lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) return 0; /* :598 (NULL = all NaN) */
    if (levelCutoffRepeat > 1 && !haveRepeated) {                 /* :601-604 */
      lvl++;
      haveRepeated = 1;
    }
  }
  memcpy(pose_io, cur, sizeof cur); /* :612-613 */
  memcpy(aff_io, aff_cur, sizeof aff_cur);
  if ((modeA != 0 && (fabsf((float)aff_io[0]) > 1.2)) || (modeB != 0 && (fabsf((float)aff_io[1]) > 200))) /* :615-617 */
    return 0;
  double rel[2];
  orc_aff_from_to(t->ref_exposure, t->exposure[0], t->ref_a, t->ref_b, aff_io[0], aff_io[1], rel);
  const float rel0 = (float)rel[0], rel1 = (float)rel[1];
  if ((modeA == 0 && (fabsf(logf(rel0)) > 1.5)) || (modeB == 0 && (fabsf(rel1) > 200))) return 0; /* :624-626 */
  if (modeA < 0) aff_io[0] = 0;
  if (modeB < 0) aff_io[1] = 0;
  return 1;
}

/* ------------------------------------------------------------------------------------ */
/* calcResScale, TrackerAndScaler.cpp:1007-1172                                          */
/* ------------------------------------------------------------------------------------ */
void orc_calc_res_scale(orc_tracker *t, int lvl, float scale, float cutoffTH, double rs[6]) {
  float E = 0;
  int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
  const int wl = t->w[lvl], hl = t->h[lvl];
  const float *dINewl = t->dIp[1][lvl];
  const float fx1l = t->fx1[lvl], fy1l = t->fy1[lvl], cx1l = t->cx1[lvl], cy1l = t->cy1[lvl];
  const float *Ki = t->Ki[lvl];
  double Rd[9];
  orc_quat_to_rot(t->T10, Rd);
  float Rf[9], M[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
  mat3f_mul(Rf, Ki, M); /* rot_f1_f0_K0_i :1022-1023 */
  const float tsl[3] = {(float)t->T10[4], (float)t->T10[5], (float)t->T10[6]};
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  const float huberTH = t->p.huber_th;
  const float maxEnergy = 2 * huberTH * cutoffTH - huberTH * huberTH;
  const int nl = t->pc_n[lvl];
  const float *lpc_u = t->pc_u[lvl], *lpc_v = t->pc_v[lvl], *lpc_idepth = t->pc_id[lvl],
              *lpc_color = t->pc_c[lvl];
  float **B = t->sb;
  t->res_evals[lvl]++;
  double E64 = 0;

  for (int i = 0; i < nl; i++) {
    const float id = lpc_idepth[i], x = lpc_u[i], y = lpc_v[i];
    float pt[3], rx[3];
    for (int r = 0; r < 3; r++) { /* :1061, :1068 */
      pt[r] = (((scale * M[r * 3]) * x + (scale * M[r * 3 + 1]) * y) + (scale * M[r * 3 + 2])) + tsl[r] * id;
      rx[r] = ((M[r * 3] * x + M[r * 3 + 1] * y) + M[r * 3 + 2]) / id;
    }
    const float u = pt[0] / pt[2];
    const float v = pt[1] / pt[2];
    const float Ku = fx1l * u + cx1l;
    const float Kv = fy1l * v + cy1l;
    const float new_idepth = id / pt[2];

    if (lvl == 0 && i % 32 == 0) { /* :1070-1100 */
      float ptT[3], ptT2[3], pt3[3];
      for (int r = 0; r < 3; r++) {
        const float kx = ((scale * Ki[r * 3]) * x + (scale * Ki[r * 3 + 1]) * y) + (scale * Ki[r * 3 + 2]);
        const float mx = ((scale * M[r * 3]) * x + (scale * M[r * 3 + 1]) * y) + (scale * M[r * 3 + 2]);
        ptT[r] = kx + tsl[r] * id;
        ptT2[r] = kx - tsl[r] * id;
        pt3[r] = mx - tsl[r] * id;
      }
      const float KuT = fx1l * (ptT[0] / ptT[2]) + cx1l, KvT = fy1l * (ptT[1] / ptT[2]) + cy1l;
      const float KuT2 = fx1l * (ptT2[0] / ptT2[2]) + cx1l, KvT2 = fy1l * (ptT2[1] / ptT2[2]) + cy1l;
      const float Ku3 = fx1l * (pt3[0] / pt3[2]) + cx1l, Kv3 = fy1l * (pt3[1] / pt3[2]) + cy1l;
      sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      sumSquaredShiftNum += 2;
    }

    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue; /* :1102 */
    const float refColor = lpc_color[i];
    float hit[3];
    interp33(dINewl, Ku, Kv, wl, hit); /* :1106 */
    if (!isfinite(hit[0])) continue;
    const float residual = hit[0] - refColor; /* :1109 */
    const float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);
    if (fabsf(residual) > cutoffTH) {
      E += maxEnergy;
      E64 += maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw);
      E64 += hw * residual * residual * (2 - hw);
      numTermsInE++;
      B[0][numTermsInWarped] = rx[0]; /* :1130-1137 */
      B[1][numTermsInWarped] = rx[1];
      B[2][numTermsInWarped] = rx[2];
      B[3][numTermsInWarped] = hit[1];
      B[4][numTermsInWarped] = hit[2];
      B[5][numTermsInWarped] = residual;
      B[6][numTermsInWarped] = hw;
      B[7][numTermsInWarped] = lpc_color[i];
      numTermsInWarped++;
    }
  }
  while (numTermsInWarped % 4 != 0) {
    for (int k = 0; k < 8; k++) B[k][numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  t->sb_n = numTermsInWarped;
  t->last_E_f64 = E64;
  rs[0] = E;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
}

/* calcGSSSEScale, TrackerAndScaler.cpp:966-1005 with ScaleAccumulator.h:34-105 */
void orc_calc_gs_scale(orc_tracker *t, int lvl, float scale, float *H_out, float *b_out) {
  if (t->use_sse) {
    orc_calc_gs_scale_sse(t, lvl, scale, H_out, b_out);
    return;
  }
  static lane_acc acc;
  acc_init(&acc, 3);
  const float fx1l = t->fx1[lvl], fy1l = t->fy1[lvl];
  const float s = scale;
  const float tx = (float)t->T10[4], ty = (float)t->T10[5], tz = (float)t->T10[6];
  float **B = t->sb;
  const int n = t->sb_n;
  t->gs_evals[lvl]++;
  for (int i = 0; i < n; i += 4) {
    float J0[4], J1[4], w[4];
    for (int k = 0; k < 4; k++) {
      const float dxfx = B[3][i + k] * fx1l;
      const float dyfy = B[4][i + k] * fy1l;
      const float rx1 = B[0][i + k], rx2 = B[1][i + k], rx3 = B[2][i + k];
      const float deno_sqrt = s * rx3 + tz;
      const float deno = 1.0f / (deno_sqrt * deno_sqrt);
      const float xno = rx1 * tz - rx3 * tx;
      const float yno = rx2 * tz - rx3 * ty;
      J0[k] = dxfx * (deno * xno) + dyfy * (deno * yno);
      J1[k] = B[5][i + k];
      w[k] = B[6][i + k];
    }
    for (int k = 0; k < 4; k++) { /* updateSSE_oneed, ScaleAccumulator.h:60-77 */
      const float J0w = J0[k] * w[k];
      acc.d[0][k] = acc.d[0][k] + J0w * J0[k];
      acc.d[1][k] = acc.d[1][k] + J0w * J1[k];
      const float J1w = J1[k] * w[k];
      acc.d[2][k] = acc.d[2][k] + J1w * J1[k];
    }
    acc.n1++;
    acc_shift(&acc, 0);
  }
  acc_shift(&acc, 1);
  const float h00 = acc_finish_entry(&acc, 0), h01 = acc_finish_entry(&acc, 1);
  *H_out = h00 * (1.0f / n); /* :1003-1004 */
  *b_out = h01 * (1.0f / n);
}

/* optimizeScale, TrackerAndScaler.cpp:854-964 */
float orc_optimize_scale(orc_tracker *t, float *scale_io, int coarsestLvl) {
  float last_residuals[ORC_MAX_LEVELS];
  for (int i = 0; i < ORC_MAX_LEVELS; i++) last_residuals[i] = NAN;
  memset(t->res_evals, 0, sizeof t->res_evals);
  memset(t->gs_evals, 0, sizeof t->gs_evals);
  const int *maxIterations = t->p.max_iterations;
  const float lambdaExtrapolationLimit = t->p.lambda_extrapolation_limit;
  const float cutoff0 = t->p.coarse_cutoff_th;
  const int fixed = t->p.fixed_schedule > 0 ? t->p.fixed_schedule : 0; /* benchmark schedule (orc_params), not the reference */
  float scale_current = *scale_io;
  int haveRepeated = 0;
  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    float H, b;
    float levelCutoffRepeat = 1;
    double resOld[6];
    orc_calc_res_scale(t, lvl, scale_current, cutoff0 * levelCutoffRepeat, resOld);
    while (!fixed && resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      orc_calc_res_scale(t, lvl, scale_current, cutoff0 * levelCutoffRepeat, resOld);
    }
    orc_calc_gs_scale(t, lvl, scale_current, &H, &b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < (fixed ? fixed : maxIterations[lvl]); iteration++) {
      float Hl = H;
      Hl *= (1 + lambda);
      float inc = -b / Hl;
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
      inc *= extrapFac;
      if (!isfinite(inc) || fabsf(inc) > scale_current) inc = 0.0; /* :907-908 */
      const float scale_new = scale_current + inc;
      double resNew[6];
      orc_calc_res_scale(t, lvl, scale_new, cutoff0 * levelCutoffRepeat, resNew);
      const int accept = fixed || (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        orc_calc_gs_scale(t, lvl, scale_new, &H, &b);
        memcpy(resOld, resNew, sizeof resOld);
        scale_current = scale_new;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      if (!fixed && !(inc > 1e-3)) break; /* :937, signed (quirk Q7) */
    }
    last_residuals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = 1;
    }
  }
  *scale_io = scale_current;
  return last_residuals[0];
}

/* ------------------------------------------------------------------------------------ */
/* FrameHessian::makeImages (UPSTREAM DSO), call sites FrontEnd.cpp:605,680              */
/* L0 = input; L(k) = 2x2 mean of L(k-1); dx,dy = central differences * 0.5 for           */
/* idx in [wl, wl*(hl-1)); non-finite gradients -> 0.  First/last row gradients are        */
/* uninitialised upstream (never read by the tracker: bounds test :786); zero here.        */
/* ------------------------------------------------------------------------------------ */
void orc_make_images(const float *image, int w, int h, int nlevels, float *const *out) {
  for (int lvl = 0; lvl < nlevels; lvl++) {
    const int wl = w >> lvl, hl = h >> lvl;
    float *d = out[lvl];
    if (lvl == 0) {
      for (int i = 0; i < wl * hl; i++) {
        d[3 * i] = image[i];
        d[3 * i + 1] = 0;
        d[3 * i + 2] = 0;
      }
    } else {
      const int wlm1 = w >> (lvl - 1);
      const float *dm = out[lvl - 1];
      for (int y = 0; y < hl; y++)
        for (int x = 0; x < wl; x++) {
          d[3 * (x + y * wl)] = 0.25f * (dm[3 * (2 * x + 2 * y * wlm1)] + dm[3 * (2 * x + 1 + 2 * y * wlm1)] +
                                         dm[3 * (2 * x + 2 * y * wlm1 + wlm1)] +
                                         dm[3 * (2 * x + 1 + 2 * y * wlm1 + wlm1)]);
          d[3 * (x + y * wl) + 1] = 0;
          d[3 * (x + y * wl) + 2] = 0;
        }
    }
    for (int idx = wl; idx < wl * (hl - 1); idx++) {
      float dx = 0.5f * (d[3 * (idx + 1)] - d[3 * (idx - 1)]);
      float dy = 0.5f * (d[3 * (idx + wl)] - d[3 * (idx - wl)]);
      if (!isfinite(dx)) dx = 0;
      if (!isfinite(dy)) dy = 0;
      d[3 * idx + 1] = dx;
      d[3 * idx + 2] = dy;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* makeCoarseDepthL0, TrackerAndScaler.cpp:143-315                                       */
/* ------------------------------------------------------------------------------------ */
void orc_make_coarse_depth_l0(orc_tracker *t, int npts, const float *pu, const float *pv,
                              const float *pidepth, const float *pweight,
                              const float *const *ref_dIp, int *n_out, float *const *pc_u,
                              float *const *pc_v, float *const *pc_idepth, float *const *pc_color) {
  const int nl = t->nlevels;
  memset(t->idepth[0], 0, sizeof(float) * t->w[0] * t->h[0]);
  memset(t->wsum[0], 0, sizeof(float) * t->w[0] * t->h[0]);
  for (int k = 0; k < npts; k++) { /* :149-164 */
    const int u = pu[k] + 0.5f;
    const int v = pv[k] + 0.5f;
    const float new_idepth = pidepth[k];
    const float weight = pweight[k];
    t->idepth[0][u + t->w[0] * v] += new_idepth * weight;
    t->wsum[0][u + t->w[0] * v] += weight;
  }
  for (int lvl = 1; lvl < nl; lvl++) { /* :166-187 */
    const int lvlm1 = lvl - 1;
    const int wl = t->w[lvl], hl = t->h[lvl], wlm1 = t->w[lvlm1];
    float *idl = t->idepth[lvl], *wsl = t->wsum[lvl];
    const float *idm = t->idepth[lvlm1], *wsm = t->wsum[lvlm1];
    for (int y = 0; y < hl; y++)
      for (int x = 0; x < wl; x++) {
        const int bidx = 2 * x + 2 * y * wlm1;
        idl[x + y * wl] = idm[bidx] + idm[bidx + 1] + idm[bidx + wlm1] + idm[bidx + wlm1 + 1];
        wsl[x + y * wl] = wsm[bidx] + wsm[bidx + 1] + wsm[bidx + wlm1] + wsm[bidx + wlm1 + 1];
      }
  }
  for (int lvl = 0; lvl < nl; lvl++) { /* dilation :190-275 */
    const int wl = t->w[lvl];
    const int wh = t->w[lvl] * t->h[lvl] - t->w[lvl];
    float *ws = t->wsum[lvl], *bak = t->wsum_bak[lvl], *idl = t->idepth[lvl];
    memcpy(bak, ws, sizeof(float) * t->w[lvl] * t->h[lvl]);
    int off[4];
    if (lvl < 2) { /* diagonal neighbours :206-225 */
      off[0] = 1 + wl;
      off[1] = -1 - wl;
      off[2] = wl - 1;
      off[3] = -wl + 1;
    } else { /* axis neighbours :249-268 */
      off[0] = 1;
      off[1] = -1;
      off[2] = wl;
      off[3] = -wl;
    }
    for (int i = wl; i < wh; i++) {
      if (bak[i] <= 0) {
        float sum = 0, num = 0, numn = 0;
        for (int k = 0; k < 4; k++)
          if (bak[i + off[k]] > 0) {
            sum += idl[i + off[k]];
            num += bak[i + off[k]];
            numn++;
          }
        if (numn > 0) {
          idl[i] = sum / numn;
          ws[i] = num / numn;
        }
      }
    }
  }
  for (int lvl = 0; lvl < nl; lvl++) { /* :278-314 */
    float *ws = t->wsum[lvl], *idl = t->idepth[lvl];
    const float *dIRefl = ref_dIp[lvl];
    const int wl = t->w[lvl], hl = t->h[lvl];
    int lpc_n = 0;
    for (int y = 2; y < hl - 2; y++)
      for (int x = 2; x < wl - 2; x++) {
        const int i = x + y * wl;
        if (ws[i] > 0) {
          idl[i] /= ws[i];
          pc_u[lvl][lpc_n] = x;
          pc_v[lvl][lpc_n] = y;
          pc_idepth[lvl][lpc_n] = idl[i];
          pc_color[lvl][lpc_n] = dIRefl[3 * i];
          if (!isfinite(pc_color[lvl][lpc_n]) || !(idl[i] > 0)) {
            idl[i] = -1;
            continue;
          }
          lpc_n++;
        } else
          idl[i] = -1;
        ws[i] = 1;
      }
    n_out[lvl] = lpc_n;
  }
}

/* ------------------------------------------------------------------------------------ */
/* ring-key search, search_place.h:25-57 (FLANN UPSTREAM replaced by exact brute force)  */
/* ------------------------------------------------------------------------------------ */
struct orc_ringdb {
  int dim, margin, k;
  float thres;
  float *keys; /* index entries, entry 0 = dummy (LoopHandler.cpp:35-39) */
  int64_t n, cap;
  float *queue;
  int64_t queue_idx;
};

/* flann::L2<float>::operator() accumulation order (groups of 4), UPSTREAM FLANN dist.h */
float orc_l2_sq(const float *a, const float *b, int dim) {
  float result = 0;
  int i = 0;
  for (; i + 3 < dim; i += 4) {
    const float d0 = a[i] - b[i], d1 = a[i + 1] - b[i + 1], d2 = a[i + 2] - b[i + 2], d3 = a[i + 3] - b[i + 3];
    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  for (; i < dim; i++) {
    const float d0 = a[i] - b[i];
    result += d0 * d0;
  }
  return result;
}

orc_ringdb *orc_ringdb_create(int dim, int margin, int k, float thres, const float *dummy) {
  orc_ringdb *db = (orc_ringdb *)xcalloc(1, sizeof *db);
  db->dim = dim;
  db->margin = margin;
  db->k = k;
  db->thres = thres;
  db->cap = 1024;
  db->keys = (float *)xcalloc((size_t)db->cap * dim, 4);
  if (dummy) memcpy(db->keys, dummy, 4 * (size_t)dim);
  db->n = 1;
  db->queue = (float *)xcalloc((size_t)margin * dim, 4);
  return db;
}
void orc_ringdb_destroy(orc_ringdb *db) {
  if (!db) return;
  free(db->keys);
  free(db->queue);
  free(db);
}
int64_t orc_ringdb_size(orc_ringdb *db) { return db->n; }
void orc_ringdb_add_points(orc_ringdb *db, const float *keys, int64_t n) {
  if (db->n + n > db->cap) {
    while (db->n + n > db->cap) db->cap *= 2;
    db->keys = (float *)realloc(db->keys, (size_t)db->cap * db->dim * 4);
    if (!db->keys) abort();
  }
  memcpy(db->keys + db->n * db->dim, keys, 4 * (size_t)n * db->dim);
  db->n += n;
}
void orc_ringdb_knn(orc_ringdb *db, const float *key, int *idx_out, float *dist_out) {
  const int k = db->k;
  for (int j = 0; j < k; j++) {
    idx_out[j] = -1;
    dist_out[j] = INFINITY;
  }
  for (int64_t i = 0; i < db->n; i++) {
    const float d = orc_l2_sq(key, db->keys + i * db->dim, db->dim);
    /* insertion keeping ascending (dist, idx): strict < keeps the smaller index first on ties */
    int pos = k;
    while (pos > 0 && d < dist_out[pos - 1]) pos--;
    if (pos < k) {
      for (int j = k - 1; j > pos; j--) {
        dist_out[j] = dist_out[j - 1];
        idx_out[j] = idx_out[j - 1];
      }
      dist_out[pos] = d;
      idx_out[pos] = (int)i;
    }
  }
}
void orc_ringdb_query_then_enqueue(orc_ringdb *db, const float *key, int *cand_out, int *ncand_out) {
  int nc = 0;
  if (db->n > db->k) { /* :29 */
    int idx[16];
    float dist[16];
    orc_ringdb_knn(db, key, idx, dist);
    for (int i = 0; i < db->k; i++)
      if (dist[i] < db->thres && idx[i] > 0) cand_out[nc++] = idx[i] - 1; /* :34-38 */
  }
  *ncand_out = nc;
  const int r_cols = db->dim; /* :41-56 */
  if (db->queue_idx >= db->margin)
    orc_ringdb_add_points(db, db->queue + (db->queue_idx % db->margin) * r_cols, 1);
  memcpy(db->queue + (db->queue_idx % db->margin) * r_cols, key, 4 * (size_t)r_cols);
  db->queue_idx++;
}

/* search_sc, search_place.h:59-84 */
float orc_sc_distance(const int *a_idx, const double *a_val, int na, const int *b_idx,
                      const double *b_val, int nb, int sc_width) {
  float cur_prod = 0;
  int m = 0, n = 0;
  while (m < na && n < nb) {
    if (a_idx[m] == b_idx[n]) {
      cur_prod += a_val[m] * b_val[n]; /* float += double*double, rounded to float each step */
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
void orc_search_sc(const int *sig_idx, const double *sig_val, int n_sig, int n_cand,
                   const int *cand_ids, const int *const *cand_idx, const double *const *cand_val,
                   const int *cand_n, int sc_width, int *res_idx, float *res_diff) {
  *res_idx = cand_ids[0];
  *res_diff = 1.1;
  for (int c = 0; c < n_cand; c++) {
    const float cur = orc_sc_distance(sig_idx, sig_val, n_sig, cand_idx[c], cand_val[c], cand_n[c], sc_width);
    if (*res_diff > cur) {
      *res_idx = cand_ids[c];
      *res_diff = cur;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* PoseEstimator ("next" row N2): src/loop_closure/pose_estimation/PoseEstimator.cpp     */
/* ------------------------------------------------------------------------------------ */
struct orc_pose_estimator {
  orc_params p;
  int nlevels, w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
  float fx[ORC_MAX_LEVELS], fy[ORC_MAX_LEVELS], cx[ORC_MAX_LEVELS], cy[ORC_MAX_LEVELS];
  int n;
  const double *xyz;            /* pts_[i].first */
  const float *const *colors;   /* pts_[i].second[lvl] as colors[lvl][i] */
  const float *const *dIp;      /* new_frame_->dIp */
  float ref_exposure, new_exposure;
  float *B[8];
  int bn;
};

orc_pose_estimator *orc_pe_create(int w, int h, int nlevels, const orc_params *p) {
  orc_pose_estimator *e = (orc_pose_estimator *)xcalloc(1, sizeof *e);
  e->p = *p;
  e->nlevels = nlevels;
  for (int l = 0; l < nlevels; l++) {
    e->w[l] = w >> l;
    e->h[l] = h >> l;
  }
  for (int i = 0; i < 8; i++) e->B[i] = (float *)xcalloc((size_t)w * h + 4, 4); /* :41-50 */
  return e;
}
void orc_pe_destroy(orc_pose_estimator *e) {
  if (!e) return;
  for (int i = 0; i < 8; i++) free(e->B[i]);
  free(e);
}

/* PoseEstimator::calcRes, :141-296 */
static void pe_calc_res(orc_pose_estimator *e, int lvl, const double pose[7], const double aff[2], float cutoffTH,
                        double rs[6]) {
  float E = 0;
  int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
  const int wl = e->w[lvl], hl = e->h[lvl];
  const float *dINewl = e->dIp[lvl];
  const float fxl = e->fx[lvl], fyl = e->fy[lvl], cxl = e->cx[lvl], cyl = e->cy[lvl];
  double Rd[9];
  orc_quat_to_rot(pose, Rd);
  float R[9];
  for (int i = 0; i < 9; i++) R[i] = (float)Rd[i]; /* :155 */
  const float t[3] = {(float)pose[4], (float)pose[5], (float)pose[6]};
  double affd[2];
  orc_aff_from_to(e->ref_exposure, e->new_exposure, 0.0, 0.0, aff[0], aff[1], affd); /* :157-160, ref_aff_g2l_ = (0,0) */
  const float affLL0 = (float)affd[0], affLL1 = (float)affd[1];
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  const float huberTH = e->p.huber_th;
  const float maxEnergy = 2 * huberTH * cutoffTH - huberTH * huberTH;
  float **B = e->B;
  for (int i = 0; i < e->n; i++) {
    const float x = e->xyz[3 * i], y = e->xyz[3 * i + 1], z = e->xyz[3 * i + 2]; /* :183-185 */
    const float u0 = x / z, v0 = y / z;
    const float Ku0 = fxl * u0 + cxl, Kv0 = fyl * v0 + cyl;
    float pt[3];
    for (int r = 0; r < 3; r++) pt[r] = ((R[r * 3] * x + R[r * 3 + 1] * y) + R[r * 3 + 2] * z) + t[r]; /* :192 */
    const float u = pt[0] / pt[2], v = pt[1] / pt[2];
    const float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
    const float new_idepth = 1 / pt[2];
    if (lvl == 0 && i % 32 == 0) { /* :199-230 */
      const float ptT[3] = {x + t[0], y + t[1], 1 + t[2]};
      const float ptT2[3] = {x - t[0], y - t[1], 1 - t[2]};
      float pt3[3];
      for (int r = 0; r < 3; r++) pt3[r] = ((R[r * 3] * x + R[r * 3 + 1] * y) + R[r * 3 + 2]) - t[r];
      const float KuT = fxl * (ptT[0] / ptT[2]) + cxl, KvT = fyl * (ptT[1] / ptT[2]) + cyl;
      const float KuT2 = fxl * (ptT2[0] / ptT2[2]) + cxl, KvT2 = fyl * (ptT2[1] / ptT2[2]) + cyl;
      const float Ku3 = fxl * (pt3[0] / pt3[2]) + cxl, Kv3 = fyl * (pt3[1] / pt3[2]) + cyl;
      sumSquaredShiftT += (KuT - Ku0) * (KuT - Ku0) + (KvT - Kv0) * (KvT - Kv0);
      sumSquaredShiftT += (KuT2 - Ku0) * (KuT2 - Ku0) + (KvT2 - Kv0) * (KvT2 - Kv0);
      sumSquaredShiftRT += (Ku - Ku0) * (Ku - Ku0) + (Kv - Kv0) * (Kv - Kv0);
      sumSquaredShiftRT += (Ku3 - Ku0) * (Ku3 - Ku0) + (Kv3 - Kv0) * (Kv3 - Kv0);
      sumSquaredShiftNum += 2;
    }
    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue; /* :235 */
    const float refColor = e->colors[lvl][i];
    float hit[3];
    interp33(dINewl, Ku, Kv, wl, hit);
    if (!isfinite(hit[0])) continue;
    const float residual = hit[0] - (float)(affLL0 * refColor + affLL1);
    const float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);
    if (fabsf(residual) > cutoffTH) {
      E += maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw);
      numTermsInE++;
      B[0][numTermsInWarped] = new_idepth;
      B[1][numTermsInWarped] = u;
      B[2][numTermsInWarped] = v;
      B[3][numTermsInWarped] = hit[1];
      B[4][numTermsInWarped] = hit[2];
      B[5][numTermsInWarped] = residual;
      B[6][numTermsInWarped] = hw;
      B[7][numTermsInWarped] = refColor;
      numTermsInWarped++;
    }
  }
  while (numTermsInWarped % 4 != 0) {
    for (int k = 0; k < 8; k++) B[k][numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  e->bn = numTermsInWarped;
  rs[0] = E;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
}

/* PoseEstimator::calcGSSSE, :84-139 (same arithmetic as calcGSSSEPose with b0 = 0) */
static void pe_calc_gs(orc_pose_estimator *e, int lvl, const double aff[2], double H_out[64], double b_out[8]) {
  static lane_acc acc;
  acc_init(&acc, 45);
  const float fxl = e->fx[lvl], fyl = e->fy[lvl];
  const float b0 = 0.0f;
  double affd[2];
  orc_aff_from_to(e->ref_exposure, e->new_exposure, 0.0, 0.0, aff[0], aff[1], affd);
  const float a = (float)affd[0];
  float **B = e->B;
  const int n = e->bn;
  for (int i = 0; i < n; i += 4) {
    float J[9][4], w[4];
    for (int k = 0; k < 4; k++) {
      const float dx = B[3][i + k] * fxl, dy = B[4][i + k] * fyl;
      const float u = B[1][i + k], v = B[2][i + k], id = B[0][i + k];
      J[0][k] = id * dx;
      J[1][k] = id * dy;
      J[2][k] = 0.0f - id * (u * dx + v * dy);
      J[3][k] = 0.0f - ((u * v) * dx + dy * (1.0f + v * v));
      J[4][k] = (u * v) * dy + dx * (1.0f + u * u);
      J[5][k] = u * dy - v * dx;
      J[6][k] = a * (b0 - B[7][i + k]);
      J[7][k] = -1.0f;
      J[8][k] = B[5][i + k];
      w[k] = B[6][i + k];
    }
    int idx = 0;
    for (int r = 0; r < 9; r++) {
      float Jw[4];
      for (int k = 0; k < 4; k++) Jw[k] = J[r][k] * w[k];
      for (int c = r; c < 9; c++) {
        for (int k = 0; k < 4; k++) acc.d[idx][k] = acc.d[idx][k] + Jw[k] * J[c][k];
        idx++;
      }
    }
    acc.n1++;
    acc_shift(&acc, 0);
  }
  acc_shift(&acc, 1);
  float Hf[9][9];
  int idx = 0;
  for (int r = 0; r < 9; r++)
    for (int c = r; c < 9; c++) {
      const float d = acc_finish_entry(&acc, idx++);
      Hf[r][c] = Hf[c][r] = d;
    }
  const float invn = 1.0f / n;
  const double s[8] = {e->p.scale_xi_rot,   e->p.scale_xi_rot,   e->p.scale_xi_rot, e->p.scale_xi_trans,
                       e->p.scale_xi_trans, e->p.scale_xi_trans, e->p.scale_a,      e->p.scale_b};
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H_out[r * 8 + c] = (((double)Hf[r][c] * (double)invn) * s[c]) * s[r];
    b_out[r] = ((double)Hf[r][8] * (double)invn) * s[r];
  }
}

/* PoseEstimator::estimate, :298-506.  Returns the reference's bool. */
int orc_pe_estimate(orc_pose_estimator *e, int n, const double *xyz, const float *const *colors, float ref_ab_exposure,
                    const float *const *new_dIp, float new_ab_exposure, const float new_cam[4], int coarsest_lvl,
                    double ref_to_new_io[16], float *pose_error, int *inlier_percent_out) {
  const int *maxIterations = e->p.max_iterations;
  const float lambdaExtrapolationLimit = e->p.lambda_extrapolation_limit;
  const float cutoff0 = e->p.coarse_cutoff_th;
  e->fx[0] = new_cam[0], e->fy[0] = new_cam[1], e->cx[0] = new_cam[2], e->cy[0] = new_cam[3]; /* makeK :62-82 */
  for (int l = 1; l < e->nlevels; l++) {
    e->fx[l] = e->fx[l - 1] * 0.5;
    e->fy[l] = e->fy[l - 1] * 0.5;
    e->cx[l] = (e->cx[0] + 0.5) / ((int)1 << l) - 0.5;
    e->cy[l] = (e->cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  e->n = n, e->xyz = xyz, e->colors = colors, e->dIp = new_dIp;
  e->ref_exposure = ref_ab_exposure, e->new_exposure = new_ab_exposure;
  int lastInners[ORC_MAX_LEVELS] = {0};
  double lastResiduals[ORC_MAX_LEVELS];
  for (int i = 0; i < ORC_MAX_LEVELS; i++) lastResiduals[i] = NAN;
  double cur[7], aff_cur[2] = {0, 0};
  orc_se3_from_matrix(ref_to_new_io, cur); /* :321-322 */
  int haveRepeated = 0;
  const float modeA = e->p.affine_opt_mode_a, modeB = e->p.affine_opt_mode_b;
  const double sc[8] = {e->p.scale_xi_rot,   e->p.scale_xi_rot,   e->p.scale_xi_rot, e->p.scale_xi_trans,
                        e->p.scale_xi_trans, e->p.scale_xi_trans, e->p.scale_a,      e->p.scale_b};
  for (int lvl = coarsest_lvl; lvl >= 0; lvl--) {
    double H[64], b[8], resOld[6];
    float levelCutoffRepeat = 1;
    pe_calc_res(e, lvl, cur, aff_cur, cutoff0 * levelCutoffRepeat, resOld);
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      pe_calc_res(e, lvl, cur, aff_cur, cutoff0 * levelCutoffRepeat, resOld);
    }
    pe_calc_gs(e, lvl, aff_cur, H, b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      double Hl[64], inc[8], nb[8];
      memcpy(Hl, H, sizeof Hl);
      for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda);
      for (int i = 0; i < 8; i++) nb[i] = -b[i];
      orc_ldlt_solve(8, Hl, nb, inc);
      if (modeA < 0 && modeB < 0) {
        double A6[36], x6[6];
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) A6[r * 6 + c] = Hl[r * 8 + c];
        orc_ldlt_solve(6, A6, nb, x6);
        for (int i = 0; i < 6; i++) inc[i] = x6[i];
        inc[6] = inc[7] = 0;
      }
      if (!(modeA < 0) && modeB < 0) {
        double A7[49], x7[7];
        for (int r = 0; r < 7; r++)
          for (int c = 0; c < 7; c++) A7[r * 7 + c] = Hl[r * 8 + c];
        orc_ldlt_solve(7, A7, nb, x7);
        for (int i = 0; i < 7; i++) inc[i] = x7[i];
        inc[7] = 0;
      }
      if (modeA < 0 && !(modeB < 0)) {
        double Hs[64], bs[8], A7[49], nbs[7], x7[7];
        memcpy(Hs, Hl, sizeof Hs);
        memcpy(bs, b, sizeof bs);
        for (int r = 0; r < 8; r++) Hs[r * 8 + 6] = Hs[r * 8 + 7];
        for (int c = 0; c < 8; c++) Hs[6 * 8 + c] = Hs[7 * 8 + c];
        bs[6] = bs[7];
        for (int r = 0; r < 7; r++) {
          for (int c = 0; c < 7; c++) A7[r * 7 + c] = Hs[r * 8 + c];
          nbs[r] = -bs[r];
        }
        orc_ldlt_solve(7, A7, nbs, x7);
        for (int i = 0; i < 8; i++) inc[i] = 0;
        for (int i = 0; i < 6; i++) inc[i] = x7[i];
        inc[7] = x7[6];
      }
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrtf(lambdaExtrapolationLimit / lambda));
      for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
      double incScaled[8], sum = 0;
      for (int i = 0; i < 8; i++) {
        incScaled[i] = inc[i] * sc[i];
        sum += incScaled[i];
      }
      if (!isfinite(sum))
        for (int i = 0; i < 8; i++) incScaled[i] = 0;
      double ex[7], newp[7], aff_new[2];
      orc_se3_exp(incScaled, ex);
      orc_se3_mul(ex, cur, newp);
      aff_new[0] = aff_cur[0] + incScaled[6];
      aff_new[1] = aff_cur[1] + incScaled[7];
      double resNew[6];
      pe_calc_res(e, lvl, newp, aff_new, cutoff0 * levelCutoffRepeat, resNew);
      const int accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        pe_calc_gs(e, lvl, aff_new, H, b);
        memcpy(resOld, resNew, sizeof resOld);
        memcpy(aff_cur, aff_new, sizeof aff_cur);
        memcpy(cur, newp, sizeof cur);
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0;
      for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      if (!(sqrt(nrm) > 1e-3)) break;
    }
    lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1])); /* :462 */
    lastInners[lvl] = resOld[1];                                /* :463 */
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = 1;
    }
  }
  { /* ref_to_new = refToNew_current.matrix(), :466 */
    double R[9];
    orc_quat_to_rot(cur, R);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) ref_to_new_io[r * 4 + c] = R[r * 3 + c];
      ref_to_new_io[r * 4 + 3] = cur[4 + r];
    }
    ref_to_new_io[12] = ref_to_new_io[13] = ref_to_new_io[14] = 0;
    ref_to_new_io[15] = 1;
  }
  *pose_error = lastResiduals[0];
  int aff_good = 1; /* :469-482 */
  if ((modeA != 0 && (fabsf((float)aff_cur[0]) > 1.2)) || (modeB != 0 && (fabsf((float)aff_cur[1]) > 200))) aff_good = 0;
  double rel[2];
  orc_aff_from_to(e->ref_exposure, e->new_exposure, 0.0, 0.0, aff_cur[0], aff_cur[1], rel);
  const float rel0 = (float)rel[0], rel1 = (float)rel[1];
  if ((modeA == 0 && (fabsf(logf(rel0)) > 1.5)) || (modeB == 0 && (fabsf(rel1) > 200))) aff_good = 0;
  const int low_res = *pose_error < 10.0; /* RES_THRES, PoseEstimator.h:26 */
  const int inlier_percent = 100 * (float)lastInners[0] / n; /* :486 */
  if (inlier_percent_out) *inlier_percent_out = inlier_percent;
  return aff_good && low_res && (inlier_percent > 90); /* INNER_PERCENT, PoseEstimator.h:27 */
}
