/*
 * dsm_oracle_internal.h -- state shared by the translation units of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see dsm_oracle.h).
 */
#ifndef DSM_ORACLE_INTERNAL_H
#define DSM_ORACLE_INTERNAL_H
#include "dsm_oracle.h"

struct orc_tracker {
  orc_params p;
  int nlevels;
  int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
  float fx[ORC_MAX_LEVELS], fy[ORC_MAX_LEVELS], cx[ORC_MAX_LEVELS], cy[ORC_MAX_LEVELS];
  float Ki[ORC_MAX_LEVELS][9];
  float fx1[ORC_MAX_LEVELS], fy1[ORC_MAX_LEVELS], cx1[ORC_MAX_LEVELS], cy1[ORC_MAX_LEVELS];
  double T10[7]; /* tfm_f1_f0_ */
  /* template */
  float *pc_u[ORC_MAX_LEVELS], *pc_v[ORC_MAX_LEVELS], *pc_id[ORC_MAX_LEVELS], *pc_c[ORC_MAX_LEVELS];
  int pc_n[ORC_MAX_LEVELS];
  float *idepth[ORC_MAX_LEVELS], *wsum[ORC_MAX_LEVELS], *wsum_bak[ORC_MAX_LEVELS];
  int ref_id;
  double ref_a, ref_b;
  float ref_exposure;
  /* frames (borrowed) */
  const float *dIp[2][ORC_MAX_LEVELS];
  float exposure[2];
  /* warped buffers: pose (idepth,u,v,dx,dy,residual,weight,refColor) */
  float *pb[8];
  int pb_n;
  /* scale (rx1,rx2,rx3,dx,dy,residual,weight,refColor) */
  float *sb[8];
  int sb_n;
  int64_t res_evals[ORC_MAX_LEVELS], gs_evals[ORC_MAX_LEVELS];
  int use_sse;       /* calcGSSSE* through the SSE-intrinsics translation unit (timed CPU baseline) */
  double last_E_f64; /* the same per-point float terms summed in double (test aid, see orc_last_energy_f64) */
};

/* dsm_oracle_sse.c: the SSE-intrinsics form of the two Gauss-Newton accumulations (the reference's own form), used
 * when t->use_sse is set (orc_tracker_use_sse); bit-identical to the scalar lane emulation in the parity build. */
void orc_calc_gs_pose_sse(orc_tracker *t, int lvl, const double aff[2], double H_out[64], double b_out[8]);
void orc_calc_gs_scale_sse(orc_tracker *t, int lvl, float scale, float *H_out, float *b_out);
void orc_aff_from_to(float expF, float expT, double g2F_a, double g2F_b, double g2T_a, double g2T_b, double out[2]);

#endif
