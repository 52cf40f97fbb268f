/*
 * dsm_oracle_sse.c -- the two Gauss-Newton accumulations of the hot path in the reference's OWN form: 4-wide SSE
 * intrinsics over the warped SoA buffers, 4-lane accumulators shifted up every 1000 packs ("1k" / "1m" buffers).
 * TEST INFRASTRUCTURE ONLY (see dsm_oracle.h): this is the TIMED CPU baseline of bench.py (SURVEY.md section 8d:
 * "the build's SSE restatement ... same loop structure and 4-lane accumulators"); the scalar lane emulation in
 * dsm_oracle.c stays the parity oracle, and tests/test_oracle_sse.py asserts the two agree bit for bit in the parity
 * build (no FMA contraction).
 *
 *   orc_calc_gs_pose_sse   TrackerAndScaler.cpp:640-697 (calcGSSSEPose) + UPSTREAM DSO Accumulator9
 *                          (initialize / updateSSE_eighted / finish; used at TrackerAndScaler.h:108, .cpp:642,664,681)
 *   orc_calc_gs_scale_sse  TrackerAndScaler.cpp:966-1005 (calcGSSSEScale) + ScaleAccumulator.h:34-105
 *
 * The residual passes (calcResPose / calcResScale, :699-852 / :1007-1172) are scalar loops in the reference too, so
 * dsm_oracle.c's restatement of them IS their timed form.
 */
#include <assert.h>
#include <stdint.h>
#include <string.h>
#include <xmmintrin.h>

#include "dsm_oracle_internal.h"

/* Accumulator with NENT 4-lane entries: the shift-up scheme of ScaleAccumulator.h:85-105 (Accumulator9 upstream uses
 * the same one).  num_in_1 counts PACKS; after the first shift num_in_1k (= 1001) already exceeds 1000, so the 1k
 * buffer is forwarded to 1m at once -- as written. */
#define ACC_MAX 45
typedef struct {
  __attribute__((aligned(16))) float d[4 * ACC_MAX];
  __attribute__((aligned(16))) float d1k[4 * ACC_MAX];
  __attribute__((aligned(16))) float d1m[4 * ACC_MAX];
  float num_in_1, num_in_1k, num_in_1m;
  int nent;
} sse_acc;

static void sse_acc_init(sse_acc *a, int nent) { /* ScaleAccumulator.h:34-41 */
  memset(a->d, 0, sizeof a->d);
  memset(a->d1k, 0, sizeof a->d1k);
  memset(a->d1m, 0, sizeof a->d1m);
  a->num_in_1 = a->num_in_1k = a->num_in_1m = 0;
  a->nent = nent;
}

static inline void sse_acc_shift_up(sse_acc *a, int force) { /* ScaleAccumulator.h:85-105 */
  if (a->num_in_1 > 1000 || force) {
    for (int i = 0; i < a->nent; i++)
      _mm_store_ps(a->d1k + 4 * i, _mm_add_ps(_mm_load_ps(a->d + 4 * i), _mm_load_ps(a->d1k + 4 * i)));
    a->num_in_1k += a->num_in_1;
    a->num_in_1 = 0;
    memset(a->d, 0, sizeof(float) * 4 * a->nent);
  }
  if (a->num_in_1k > 1000 || force) {
    for (int i = 0; i < a->nent; i++)
      _mm_store_ps(a->d1m + 4 * i, _mm_add_ps(_mm_load_ps(a->d1k + 4 * i), _mm_load_ps(a->d1m + 4 * i)));
    a->num_in_1m += a->num_in_1k;
    a->num_in_1k = 0;
    memset(a->d1k, 0, sizeof(float) * 4 * a->nent);
  }
}

static inline float sse_acc_entry(const sse_acc *a, int idx) { /* finish(): lanes summed last, ScaleAccumulator.h:51-53 */
  const float *p = a->d1m + 4 * idx;
  return p[0] + p[1] + p[2] + p[3];
}

/* Accumulator9::updateSSE_eighted (UPSTREAM DSO): H(r,c) += (J_r * w) * J_c over the upper triangle, r-major */
static inline void acc9_update_weighted(sse_acc *a, const __m128 J[9], const __m128 w) {
  float *pt = a->d;
  for (int r = 0; r < 9; r++) {
    const __m128 Jw = _mm_mul_ps(J[r], w);
    for (int c = r; c < 9; c++) {
      _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(Jw, J[c])));
      pt += 4;
    }
  }
  a->num_in_1++;
  sse_acc_shift_up(a, 0);
}

void orc_calc_gs_pose_sse(orc_tracker *t, int lvl, const double aff[2], double H_out[64], double b_out[8]) {
  static sse_acc acc; /* single-threaded test infrastructure (one per forked worker in the all-core leg) */
  sse_acc_init(&acc, 45); /* :642 */
  const __m128 fxl = _mm_set1_ps(t->fx[lvl]); /* :644-645 */
  const __m128 fyl = _mm_set1_ps(t->fy[lvl]);
  const __m128 b0 = _mm_set1_ps((float)t->ref_b); /* :646 */
  double affd[2];
  orc_aff_from_to(t->ref_exposure, t->exposure[0], t->ref_a, t->ref_b, aff[0], aff[1], affd);
  const __m128 a = _mm_set1_ps((float)affd[0]); /* :647-649 */
  const __m128 one = _mm_set1_ps(1), minusOne = _mm_set1_ps(-1), zero = _mm_set1_ps(0);
  float *const *B = t->pb; /* idepth,u,v,dx,dy,residual,weight,refColor */
  const int n = t->pb_n;
  assert(n % 4 == 0); /* :656 */
  for (int k = 0; k < 8; k++) assert(((uintptr_t)B[k] & 15) == 0);
  t->gs_evals[lvl]++;
  for (int i = 0; i < n; i += 4) {
    const __m128 dx = _mm_mul_ps(_mm_load_ps(B[3] + i), fxl); /* :658-662 */
    const __m128 dy = _mm_mul_ps(_mm_load_ps(B[4] + i), fyl);
    const __m128 u = _mm_load_ps(B[1] + i);
    const __m128 v = _mm_load_ps(B[2] + i);
    const __m128 id = _mm_load_ps(B[0] + i);
    __m128 J[9];
    J[0] = _mm_mul_ps(id, dx); /* :664-678 */
    J[1] = _mm_mul_ps(id, dy);
    J[2] = _mm_sub_ps(zero, _mm_mul_ps(id, _mm_add_ps(_mm_mul_ps(u, dx), _mm_mul_ps(v, dy))));
    J[3] = _mm_sub_ps(zero, _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dx), _mm_mul_ps(dy, _mm_add_ps(one, _mm_mul_ps(v, v)))));
    J[4] = _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dy), _mm_mul_ps(dx, _mm_add_ps(one, _mm_mul_ps(u, u))));
    J[5] = _mm_sub_ps(_mm_mul_ps(u, dy), _mm_mul_ps(v, dx));
    J[6] = _mm_mul_ps(a, _mm_sub_ps(b0, _mm_load_ps(B[7] + i)));
    J[7] = minusOne;
    J[8] = _mm_load_ps(B[5] + i);
    acc9_update_weighted(&acc, J, _mm_load_ps(B[6] + i));
  }
  sse_acc_shift_up(&acc, 1); /* finish(), :681 */
  float Hf[9][9];
  int idx = 0;
  for (int r = 0; r < 9; r++)
    for (int c = r; c < 9; c++) {
      const float d = sse_acc_entry(&acc, idx++);
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

void orc_calc_gs_scale_sse(orc_tracker *t, int lvl, float scale, float *H_out, float *b_out) {
  static sse_acc acc;
  sse_acc_init(&acc, 3); /* :968 */
  const __m128 fx1l = _mm_set1_ps(t->fx1[lvl]); /* :970-971 */
  const __m128 fy1l = _mm_set1_ps(t->fy1[lvl]);
  const __m128 s = _mm_set1_ps(scale); /* :973-976 */
  const __m128 tx = _mm_set1_ps((float)t->T10[4]);
  const __m128 ty = _mm_set1_ps((float)t->T10[5]);
  const __m128 tz = _mm_set1_ps((float)t->T10[6]);
  const __m128 one = _mm_set1_ps(1);
  float *const *B = t->sb; /* rx1,rx2,rx3,dx,dy,residual,weight,refColor */
  const int n = t->sb_n;
  assert(n % 4 == 0); /* :981 */
  t->gs_evals[lvl]++;
  for (int i = 0; i < n; i += 4) {
    const __m128 dxfx = _mm_mul_ps(_mm_load_ps(B[3] + i), fx1l); /* :983-987 */
    const __m128 dyfy = _mm_mul_ps(_mm_load_ps(B[4] + i), fy1l);
    const __m128 rx1 = _mm_load_ps(B[0] + i);
    const __m128 rx2 = _mm_load_ps(B[1] + i);
    const __m128 rx3 = _mm_load_ps(B[2] + i);
    const __m128 deno_sqrt = _mm_add_ps(_mm_mul_ps(s, rx3), tz); /* :989-990 */
    const __m128 deno = _mm_div_ps(one, _mm_mul_ps(deno_sqrt, deno_sqrt));
    const __m128 xno = _mm_sub_ps(_mm_mul_ps(rx1, tz), _mm_mul_ps(rx3, tx)); /* :992-993 */
    const __m128 yno = _mm_sub_ps(_mm_mul_ps(rx2, tz), _mm_mul_ps(rx3, ty));
    /* updateSSE_oneed, ScaleAccumulator.h:60-77 */
    const __m128 J0 = _mm_add_ps(_mm_mul_ps(dxfx, _mm_mul_ps(deno, xno)), _mm_mul_ps(dyfy, _mm_mul_ps(deno, yno)));
    const __m128 J1 = _mm_load_ps(B[5] + i);
    const __m128 w = _mm_load_ps(B[6] + i);
    float *pt = acc.d;
    const __m128 J0w = _mm_mul_ps(J0, w);
    _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(J0w, J0)));
    pt += 4;
    _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(J0w, J1)));
    pt += 4;
    const __m128 J1w = _mm_mul_ps(J1, w);
    _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(J1w, J1)));
    acc.num_in_1++;
    sse_acc_shift_up(&acc, 0);
  }
  sse_acc_shift_up(&acc, 1); /* finish(), :1002 */
  *H_out = sse_acc_entry(&acc, 0) * (1.0f / n); /* :1003-1004 */
  *b_out = sse_acc_entry(&acc, 1) * (1.0f / n);
}
