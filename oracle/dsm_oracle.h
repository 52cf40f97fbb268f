/*
 * dsm_oracle.h -- CPU restatement ("oracle") of the direct photometric hot path of
 * IRVLab/direct_stereo_slam.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (direct_stereo_slam_amd/, include/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors and
 * cannot be compiled in this image (every hot-path translation unit includes Eigen, Sophus
 * and the un-vendored libdso headers; dependencies.zip is absent -- .MISSING_LARGE_BLOBS:1).
 * The oracle is therefore a line-by-line restatement of the reference sources cited at each
 * function, plus restatements of the published algorithms of the absent third-party pieces
 * (DSO Accumulator9 / getInterpolatedElement33 / AffLight / makeImages, Sophus SE3, Eigen
 * LDLT, FLANN L2 k-NN).  It is cross-checked against independent numpy/scipy computations in
 * tests/test_oracle_*.py, never against outputs of the reference binary.
 */
#ifndef DSM_ORACLE_H
#define DSM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 6

typedef struct orc_params {
  float huber_th, coarse_cutoff_th;
  float scale_xi_rot, scale_xi_trans, scale_a, scale_b;
  float affine_opt_mode_a, affine_opt_mode_b;
  float lambda_extrapolation_limit;
  int max_iterations[ORC_MAX_LEVELS];
  int fixed_schedule; /* K > 0: the BENCHMARK schedule of SURVEY.md section 8d (1 + K evaluations per level, every step taken:
                         no accept test, cut-off repeat, small-step break or abort) -- not the reference's algorithm; 0 = as written */
} orc_params;

typedef struct orc_tracker orc_tracker;
typedef struct orc_ringdb orc_ringdb;

void orc_params_default(orc_params *p);

orc_tracker *orc_tracker_create(int w, int h, int nlevels, const double T_f1_f0[16],
                                const float K1[4], const orc_params *p);
void orc_tracker_destroy(orc_tracker *t);
void orc_tracker_make_k(orc_tracker *t, float fx, float fy, float cx, float cy);
void orc_tracker_set_ref(orc_tracker *t, int ref_id, double ref_a, double ref_b, float ref_exposure,
                         const int *n, const float *const *pc_u, const float *const *pc_v,
                         const float *const *pc_idepth, const float *const *pc_color);
void orc_tracker_scale_depth(orc_tracker *t, float scale);
int orc_tracker_get_template(orc_tracker *t, int lvl, int *n, float *u, float *v, float *id, float *c);
/* slot 0 = new left frame, 1 = right frame (fh1_).  dIp pointers are BORROWED (as in the reference). */
void orc_tracker_set_frame(orc_tracker *t, int slot, const float *const *dIp, float ab_exposure);

/* calcResPose, TrackerAndScaler.cpp:699-852 */
void orc_calc_res_pose(orc_tracker *t, int lvl, const double pose[7], const double aff[2],
                       float cutoff_th, double rs[6]);
/* calcGSSSEPose, TrackerAndScaler.cpp:640-697 (consumes the buffers of the last calcResPose) */
void orc_calc_gs_pose(orc_tracker *t, int lvl, const double pose[7], const double aff[2],
                      double H[64], double b[8]);
int orc_pose_warped_n(orc_tracker *t);
/* on != 0: calcGSSSEPose / calcGSSSEScale run in their SSE-intrinsics form (dsm_oracle_sse.c, the reference's own
 * form and the TIMED CPU baseline of bench.py); 0 (default): the scalar lane emulation (the parity oracle).  Same
 * results bit for bit in the parity build. */
void orc_tracker_use_sse(orc_tracker *t, int on);
/* test aid: the energy of the last calcRes* with the SAME per-point float terms summed in double.
 * The reference accumulates E in float in point order (quirk Q1); with 1e5..5e5 terms that sum
 * carries a relative error up to ~n*2^-24, far above the device's tree reduction error. */
double orc_last_energy_f64(orc_tracker *t);
/* trackNewestCoarse, TrackerAndScaler.cpp:451-638.  Returns the reference's bool. */
int orc_track(orc_tracker *t, double pose_io[7], double aff_io[2], int coarsest_lvl,
              const double *min_res_for_abort, double *last_residuals, double flow_out[3]);
/* per-call counters of the last orc_track / orc_optimize_scale: res/gs evaluations per level */
void orc_get_eval_counts(orc_tracker *t, int64_t res_evals[ORC_MAX_LEVELS], int64_t gs_evals[ORC_MAX_LEVELS]);

/* calcResScale :1007-1172, calcGSSSEScale :966-1005, optimizeScale :854-964 */
void orc_calc_res_scale(orc_tracker *t, int lvl, float scale, float cutoff_th, double rs[6]);
void orc_calc_gs_scale(orc_tracker *t, int lvl, float scale, float *H, float *b);
int orc_scale_warped_n(orc_tracker *t);
float orc_optimize_scale(orc_tracker *t, float *scale_io, int coarsest_lvl);

/* upstream DSO FrameHessian::makeImages (call sites FrontEnd.cpp:605,680): float image ->
 * per-level AoS (I,dx,dy).  out[lvl] must hold 3*w_l*h_l floats. */
void orc_make_images(const float *image, int w, int h, int nlevels, float *const *out);

/* makeCoarseDepthL0, TrackerAndScaler.cpp:143-315, from a flat list of active-point
 * projections (u, v, idepth, weight) instead of the PointHessian graph.  ref_dIp = the
 * reference frame's pyramid.  Outputs per-level lists (capacity w_l*h_l each). */
void orc_make_coarse_depth_l0(orc_tracker *t, int npts, const float *pu, const float *pv,
                              const float *pidepth, const float *pweight,
                              const float *const *ref_dIp, int *n_out, float *const *pc_u,
                              float *const *pc_v, float *const *pc_idepth, float *const *pc_color);

/* PoseEstimator (src/loop_closure/pose_estimation/PoseEstimator.cpp): calcRes :141-296, calcGSSSE :84-139,
 * estimate :298-506.  colors[lvl][i] = pts_[i].second[lvl]. */
typedef struct orc_pose_estimator orc_pose_estimator;
orc_pose_estimator *orc_pe_create(int w, int h, int nlevels, const orc_params *p);
void orc_pe_destroy(orc_pose_estimator *e);
int orc_pe_estimate(orc_pose_estimator *e, int n, const double *xyz, const float *const *colors,
                    float ref_ab_exposure, const float *const *new_dIp, float new_ab_exposure,
                    const float new_cam[4], int coarsest_lvl, double ref_to_new_io[16], float *pose_error,
                    int *inlier_percent_out);

/* Sophus restatements exposed for property tests */
void orc_se3_exp(const double xi[6], double pose_out[7]);
void orc_se3_mul(const double a[7], const double b[7], double out[7]);
void orc_quat_to_rot(const double q[4], double R[9]);
void orc_se3_from_matrix(const double T[16], double pose_out[7]);
/* Eigen LDLT (pivoted, lower) solve of the n x n row-major symmetric system A x = rhs */
void orc_ldlt_solve(int n, const double *A, const double *rhs, double *x);

/* ring-key DB: exact brute-force replacement of the FLANN index + delay queue of
 * search_ringkey (search_place.h:25-57).  Tie-break: smaller index first. */
orc_ringdb *orc_ringdb_create(int dim, int margin, int k, float thres, const float *dummy_key);
void orc_ringdb_destroy(orc_ringdb *db);
int64_t orc_ringdb_size(orc_ringdb *db);
void orc_ringdb_add_points(orc_ringdb *db, const float *keys, int64_t n);
void orc_ringdb_query_then_enqueue(orc_ringdb *db, const float *key, int *cand_out, int *ncand_out);
/* raw k-NN (no threshold): idx_out/dist_out hold k entries, -1 / inf when fewer exist */
void orc_ringdb_knn(orc_ringdb *db, const float *key, int *idx_out, float *dist_out);
float orc_l2_sq(const float *a, const float *b, int dim);

/* search_sc inner loop, search_place.h:67-79, and the arg-min over candidates :59-84 */
float orc_sc_distance(const int *a_idx, const double *a_val, int na, const int *b_idx,
                      const double *b_val, int nb, int sc_width);
void orc_search_sc(const int *sig_idx, const double *sig_val, int n_sig, int n_cand,
                   const int *cand_ids, const int *const *cand_idx, const double *const *cand_val,
                   const int *cand_n, int sc_width, int *res_idx, float *res_diff);

#ifdef __cplusplus
}
#endif
#endif
