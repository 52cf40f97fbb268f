// dsm_device.hpp -- device-visible data structures of the MI355X hot path.
//
// Data layout in HBM:
//   template  : per level one float4 {u, v, idepth, color} per point (16 B, one coalesced
//               global_load_dwordx4 per lane)           <- pc_u/pc_v/pc_idepth/pc_color SoA of the
//               reference (TrackerAndScaler.h:90-94), interleaved at upload time
//   target    : per level the intensity plane (channel 0 of the reference's AoS (I,dx,dy) texels FrameHessian::dIp,
//               TrackerAndScaler.cpp:709,1016), 4 B per texel, row-major; gradients are formed from neighbours
//   partials  : per problem, per chunk 64 floats (45 upper-triangular 9x9 sums, E, flow sums,
//               integer counts)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DSM_MAX_LEVELS 6

namespace dsm {

constexpr int kThreads = 256;      // workgroup size of the eval kernels (4 waves)
constexpr int kPartialStride = 64; // floats per chunk partial
constexpr int kNumAcc = 45;        // upper triangle of the 9x9 accumulator (Accumulator9)
// partial slot indices
constexpr int kSlotE = 45, kSlotFlowT = 46, kSlotFlowRT = 47, kSlotFlowNum = 48;
constexpr int kSlotNTerms = 49, kSlotNSat = 50, kSlotNWarped = 51;
constexpr int kNumSlots = 52;

constexpr int kTexel = 1;         // floats per target texel: the intensity; gradients are formed from neighbours (taps_interp)

struct LevelDev {
  const float4 *pts;   // n template points
  const float *img[2]; // slot 0 = new left frame, slot 1 = right frame: intensity planes
  int n, w, h, pad;
  float fx, fy, cx, cy;     // camera 0 (makeK, TrackerAndScaler.cpp:117-133)
  float Ki[9];              // inverse of K at this level (float, :135-140)
  float fx1, fy1, cx1, cy1; // camera 1 (:89-98)
};

struct ParamsDev {
  float huber_th, coarse_cutoff_th;
  float scale_xi_rot, scale_xi_trans, scale_a, scale_b;
  float affine_opt_mode_a, affine_opt_mode_b;
  float lambda_extrapolation_limit;
  int max_iterations[DSM_MAX_LEVELS];
  int fixed_schedule; // dsm_params.fixed_schedule: K > 0 = benchmark schedule (1 + K evaluations per level, every step taken)
  int geometry;       // dsm_params.chunk_geometry: which table picks the points per thread of a chunk (dsm_kernels.hpp)
};

// Read-only (during track / optimize_scale) description of one TrackerAndScaler.
struct alignas(16) TrackerDev {
  LevelDev lv[DSM_MAX_LEVELS];
  ParamsDev p;
  int nlevels;
  int pad0;
  double ref_a, ref_b; // lastRef_aff_g2l
  float ref_exposure;  // lastRef->ab_exposure
  float exposure[2];   // new_frame_->ab_exposure, fh1_->ab_exposure
  double T10[7];       // tfm_f1_f0_ as {qx,qy,qz,qw,tx,ty,tz}
};

// Inputs of one fused evaluation, produced on the device by the LM kernel.
struct EvalIn {
  // level data copied from the tracker descriptor so that an eval workgroup needs ONE dependent
  // (scalar) read of this struct instead of chasing tracker -> level -> pointer
  const float4 *pts;
  const float *img;
  int n, w, h;
  int ppt, pad0, pad1, pad2; // points per thread of a chunk of this evaluation (pts_per_thread of n under the tracker's table); the struct stays a multiple of 16 bytes
  int residual_only; // the LM loop ends after this evaluation whatever it yields (:588, the iteration bound): only the
                     // residual side (calcResPose / calcResScale) is needed -- the normal equations calcGSSSE* would build
                     // from it are never read by the reference either
  float fx, fy, cx, cy; // intrinsics of the camera the points are projected into (cam0 pose / cam1 scale)
  float Ki[9];          // K^-1 of camera 0 at this level (flow indicators)
  float huber;
  float M[9];   // pose: R*Ki ("RKi", :715) ; scale: R10*Ki ("rot_f1_f0_K0_i", :1022)
  float t[3];   // pose: translation (:716) ; scale: tsl_f1_f0 (:1024)
  float aff0, aff1; // affLL (:717-720) (pose only)
  float b0;     // (float) lastRef_aff_g2l.b (:646) (pose only)
  float scale;  // scale only
  float cutoff; // setting_coarseCutoffTH * levelCutoffRepeat
  float max_energy; // :726-728
};

enum LMPhase { PH_INIT = 0, PH_ITER = 1 };
enum LMStatus { ST_IDLE = 0, ST_RUNNING = 1, ST_GOOD = 2, ST_ABORTED = 3, ST_BAD_AFFINE = 4 };

// Per-problem state of the Levenberg-Marquardt driver (trackNewestCoarse :451-638 /
// optimizeScale :854-964), resident in device memory for the whole call.
struct alignas(16) LMState {
  int status, lvl, phase, iteration;
  int have_repeated, coarsest, is_scale /* problem kind: 0 pose, 1 scale, 2 loop-closure pose (3-D points) */;
  int stepped; // tick engine: this tick's LM step(s) of the slot already ran inside the evaluation launch (a chain, tracker_kernels.hip);
               // the tick's LM launch clears the flag and leaves the slot alone
  float lambda, level_cutoff_repeat;
  float inc_f;      // scale: last increment (for the signed break test :937)
  float scale_cur, scale_cand;
  float Hs, bs;     // scale: H, b
  int n0;           // template points of level 0 (lm_start_problem): the tick engine's default chains semi-dense problems only
  double inc_norm;  // pose: |inc| of the proposal being evaluated (:588)
  double cur[7], aff_cur[2];
  double cand[7], aff_cand[2];
  double H[64], b[8];
  double res_old[6];
  double last_residuals[DSM_MAX_LEVELS];
  double last_inners[DSM_MAX_LEVELS]; // numTermsInE at the end of each level (PoseEstimator.cpp:463)
  double min_res[DSM_MAX_LEVELS];
  double flow[3];
  long long evals[DSM_MAX_LEVELS];
  long long evals_ro[DSM_MAX_LEVELS]; // ... of which residual-only (EvalIn::residual_only)
  long long rounds[DSM_MAX_LEVELS]; // LM steps taken at each level = evaluation launches it needed (with a speculative
                                    // candidate consumed a step covers two evaluations)
  EvalIn in;
  // Speculative second candidate (dsm_params.speculate): next to the proposal being evaluated, the proposal that WOULD follow
  // if this one is rejected (same H, b and current pose; lambda four times larger, :583-585) is evaluated in the same launch.
  // A rejection -- half of all steps on the coarse levels -- then finds its successor's residual already computed and the LM
  // step consumes both: same evaluations, same decisions, same counts as the sequential loop, fewer launches in a row.
  int spec_valid;
  int ticks; // tick engine: ticks this problem has lived (= LM rounds, less the rounds a chain ran inside one tick)
  float spec_scale_cand, spec_inc_f;
  double spec_inc_norm;
  double spec_cand[7], spec_aff_cand[2];
  EvalIn spec_in;
};

// Output of a single fused evaluation (dsm_tracker_calc_res_pose / _scale)
struct SingleOut {
  double rs[6];
  double H[64];
  double b[8];
  float Hs, bs;
  int n_warped, pad;
};

} // namespace dsm
