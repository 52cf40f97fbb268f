// loopdet_kernels.hip -- device form of the two per-keyframe loops that feed the ring-key search (SURVEY.md section 8f row
// N3, second half), batched over many keyframes (one per concurrent sequence):
//
//   generate_spherical_points  (src/loop_closure/loop_detection/generate_spherical_points.h:44-85; call site
//                               LoopHandler.cpp:186-187): transform the nearby points into the current camera frame, drop
//                               the ones at or beyond lidar_range, keep per 1 x 0.5 x 1 m voxel the HIGHEST point
//                               (smallest y; the first one on ties);
//   ScanContext::generate      (ScanContext.cpp:96-141 after align_points_PCA :19-66): polar binning of the PCA-aligned
//                               points, per-bin maximum height, ring key = occupied sectors per ring / num_s, sparse
//                               signature normalised per sector.
//
// Device design.  The voxel map (the reference's unordered_map<int, pair<int, Vector3d>>) is a dense grid of
// voxel_size[0..2] cells per job -- 1.06 M cells at lidar_range 40 -- so "highest point per voxel" is two atomic-min passes
// (smallest y as order-preserving 64-bit keys, then the smallest point index among the points that hold it: exactly the
// reference's "first one wins ties") and the output order, ascending voxel index, is a stream compaction of the grid: no
// sort, no hash collisions, deterministic.  The polar bins are 1200 atomic-max cells.  Every per-point value is computed
// with the host form's operation order in double precision (-ffp-contract=off), the PCA moments are summed in point order
// by one lane per moment, the 3x3 eigen-decomposition runs the host form's cyclic Jacobi (one function for both,
// loopdet_internal.hpp) in sc_pca_kernel between the two halves -- nothing leaves the device mid-batch: results are
// bit-identical to dsm_generate_spherical_points / dsm_scancontext_generate wherever libm agrees (atan2, within 1 ulp of a
// sector boundary, is the one place it may not).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dsm_internal.hpp"
#include "loopdet_internal.hpp"

using namespace dsm;

namespace {

int invalid(const char *m) {
  set_error(m);
  return DSM_ERR_INVALID;
}

constexpr int kLdThreads = 256;
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kEmptyIdx = 0x7FFFFFFF;

// order-preserving map double -> uint64 (finite values; -0.0 is canonicalised to +0.0 first so that it ties with +0.0 as
// the host's `<` does)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  v = v + 0.0;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

struct JobDev {             // one keyframe
  const double *xyz;        // n_pts x 3 world coordinates
  const unsigned char *keep; // n_pts: the owning keyframe survived the orientation trim (may be null: all kept)
  int n_pts;
  double cw[12];            // camera <- world, row-major 3x4
  // outputs / workspace of the voxel filter
  unsigned long long *grid_y; // cells
  int *grid_idx;              // cells
  int *block_count;           // ceil(cells / kLdThreads)
  int *sel_idx;               // n_pts capacity
  double *sph;                // n_pts x 3 capacity
  int *n_out;
  // ScanContext
  double *moments;            // [0..2] mean, [3..8] cov (xx, xy, xz, yy, yz, zz)
  double mean[3], V[9];       // sc_pca_kernel: centroid and eigenvectors (ScanContext.cpp:41-47)
  double tfm[16];             // ... and tfm_pca_rig (:55-64)
  unsigned long long *bins;   // num_s * num_r order-preserving keys of max_height
  float *ringkey;             // num_r
  int *sig_idx;               // num_s * num_r capacity
  double *sig_val;
  int *n_sig;
};

// p_l = cur_cw.matrix3x4() * (g, 1)  (:57-58), same operation order as the host form
__device__ __forceinline__ void to_camera(const double *cw, const double *g, double p[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) p[r] = ((cw[r * 4 + 0] * g[0] + cw[r * 4 + 1] * g[1]) + cw[r * 4 + 2] * g[2]) + cw[r * 4 + 3] * 1.0;
}

// in range and kept?  returns the voxel cell or -1  (:53-70)
__device__ __forceinline__ long long voxel_of(const JobDev &J, int i, double lidar_range, long long vs0, long long vs1, double p[3]) {
  if (J.keep && !J.keep[i]) return -1;
  to_camera(J.cw, J.xyz + 3 * (size_t)i, p);
  if (sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) >= lidar_range) return -1;
  const long long xi = (long long)floor((p[0] + lidar_range) * 1.0), yi = (long long)floor((p[1] + lidar_range) * 2.0),
                  zi = (long long)floor((p[2] + lidar_range) * 1.0); // steps 1/RES_X, 1/RES_Y, 1/RES_Z (:23-25,:44)
  return xi + yi * vs0 + zi * vs0 * vs1;
}

__global__ void voxel_clear_kernel(const JobDev *jobs, long long cells) {
  const JobDev &J = jobs[blockIdx.y];
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (long long)gridDim.x * blockDim.x) {
    J.grid_y[c] = kEmptyKey;
    J.grid_idx[c] = kEmptyIdx;
  }
}
// pass 1: smallest y per voxel
__global__ void voxel_min_y_kernel(const JobDev *jobs, double lidar_range, long long vs0, long long vs1) {
  const JobDev &J = jobs[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < J.n_pts; i += gridDim.x * blockDim.x) {
    double p[3];
    const long long c = voxel_of(J, i, lidar_range, vs0, vs1, p);
    if (c >= 0) atomicMin(&J.grid_y[c], ordered_key(p[1]));
  }
}
// pass 2: among the points that hold their voxel's smallest y, the smallest index ("first one wins")
__global__ void voxel_min_idx_kernel(const JobDev *jobs, double lidar_range, long long vs0, long long vs1) {
  const JobDev &J = jobs[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < J.n_pts; i += gridDim.x * blockDim.x) {
    double p[3];
    const long long c = voxel_of(J, i, lidar_range, vs0, vs1, p);
    if (c >= 0 && J.grid_y[c] == ordered_key(p[1])) atomicMin(&J.grid_idx[c], i);
  }
}
// compaction in ascending voxel order: occupied cells per block of kLdThreads cells ...
__global__ void voxel_count_kernel(const JobDev *jobs, long long cells) {
  const JobDev &J = jobs[blockIdx.y];
  const long long c = (long long)blockIdx.x * kLdThreads + threadIdx.x;
  const bool occ = c < cells && J.grid_idx[c] != kEmptyIdx;
  const int n = __syncthreads_count(occ);
  if (threadIdx.x == 0) J.block_count[blockIdx.x] = n;
}
// ... exclusive scan of the block counts by one workgroup per job ...
__global__ void voxel_scan_kernel(const JobDev *jobs, int nblocks) {
  const JobDev &J = jobs[blockIdx.x];
  __shared__ int part[kLdThreads];
  const int per = (nblocks + kLdThreads - 1) / kLdThreads;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; b++) s += J.block_count[b];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < kLdThreads; t++) {
      const int v = part[t];
      part[t] = run;
      run += v;
    }
    *J.n_out = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int b = b0; b < b1; b++) {
    const int v = J.block_count[b];
    J.block_count[b] = run;
    run += v;
  }
}
// ... and the ordered write of (index, point): the point is recomputed from its world coordinates (same bits as pass 1)
__global__ void voxel_emit_kernel(const JobDev *jobs, long long cells) {
  const JobDev &J = jobs[blockIdx.y];
  const long long c = (long long)blockIdx.x * kLdThreads + threadIdx.x;
  const int idx = c < cells ? J.grid_idx[c] : kEmptyIdx;
  const bool occ = idx != kEmptyIdx;
  __shared__ int wave_base[kLdThreads / 64];
  const unsigned long long m = __ballot(occ);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_base[wave] = __popcll(m);
  __syncthreads();
  int base = J.block_count[blockIdx.x];
  for (int w = 0; w < wave; w++) base += wave_base[w];
  if (occ) {
    const int o = base + __popcll(m & ((1ull << lane) - 1ull));
    double p[3];
    to_camera(J.cw, J.xyz + 3 * (size_t)idx, p);
    J.sel_idx[o] = idx;
    J.sph[3 * (size_t)o + 0] = p[0], J.sph[3 * (size_t)o + 1] = p[1], J.sph[3 * (size_t)o + 2] = p[2];
  }
}

// ---- ScanContext ----------------------------------------------------------------------------------------------------------
// align_points_PCA :22-40: mean (sum in point order / n), then the covariance sums of the centred points in point order; one
// lane per moment, so the sums have the host form's order and bits
__global__ void sc_moments_kernel(const JobDev *jobs) {
  const JobDev &J = jobs[blockIdx.x];
  const int n = *J.n_out, t = threadIdx.x;
  __shared__ double mean[3];
  if (t < 3) {
    double s = 0;
    for (int i = 0; i < n; i++) s += J.sph[3 * (size_t)i + t];
    mean[t] = n > 0 ? s / n : 0.0;
    J.moments[t] = mean[t];
  }
  __syncthreads();
  if (t < 6) {
    const int a = t < 3 ? 0 : t < 5 ? 1 : 2, b = t < 3 ? t : t < 5 ? t - 2 : 2; // xx xy xz yy yz zz
    double s = 0;
    for (int i = 0; i < n; i++) s += (J.sph[3 * (size_t)i + a] - mean[a]) * (J.sph[3 * (size_t)i + b] - mean[b]);
    J.moments[3 + t] = s;
  }
}
__global__ void sc_clear_kernel(const JobDev *jobs, int nbins, double lidar_range) {
  const JobDev &J = jobs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nbins) J.bins[i] = ordered_key(-lidar_range - 1.0); // :93-94
}
// :96-119
__global__ void sc_bin_kernel(const JobDev *jobs, double lidar_range, int num_s, int num_r) {
  const JobDev &J = jobs[blockIdx.y];
  const int n = *J.n_out;
  const double *V = J.V;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = J.sph[3 * (size_t)i] - J.mean[0], y = J.sph[3 * (size_t)i + 1] - J.mean[1], z = J.sph[3 * (size_t)i + 2] - J.mean[2];
    const double xp = x * V[0] + y * V[3] + z * V[6]; // pts_mat * v0  (x: up)
    const double yp = x * V[1] + y * V[4] + z * V[7];
    const double zp = x * V[2] + y * V[5] + z * V[8];
    const double rho = sqrt(yp * yp + zp * zp);
    double theta = atan2(zp, yp);
    while (theta < 0) theta += 2.0 * M_PI;
    while (theta >= 2.0 * M_PI) theta -= 2.0 * M_PI;
    const int si = theta / (2.0 * M_PI) * num_s;
    const int ri = rho / lidar_range * num_r;
    if (ri >= num_r || si >= num_s) continue; // :113-114; si: the reference only asserts
    atomicMax(&J.bins[(size_t)si * num_r + ri], ordered_key(xp));
  }
}
// :122-141 by one workgroup per job: ring key, signature in ascending bin order, per-sector norms summed in ring order
__global__ void sc_finish_kernel(const JobDev *jobs, double lidar_range, int num_s, int num_r) {
  const JobDev &J = jobs[blockIdx.x];
  const int t = threadIdx.x, nbins = num_s * num_r;
  extern __shared__ double sh_norm[]; // num_s
  const unsigned long long thres = ordered_key(-lidar_range);
  if (t < num_r) { // ringkey[r] = (number of sectors whose bin (s, r) is occupied) / num_s, float adds of 1.0f as at :125
    float c = 0.0f;
    for (int s = 0; s < num_s; s++)
      if (J.bins[(size_t)s * num_r + t] >= thres) c += 1.0f;
    J.ringkey[t] = c / num_s;
  }
  if (t < num_s) { // sig_norm_si(s) += h*h over the sector's rings in ascending bin order (:127)
    double s2 = 0;
    for (int r = 0; r < num_r; r++) {
      const unsigned long long k = J.bins[(size_t)t * num_r + r];
      if (k >= thres) {
        const double h = key_to_double(k);
        s2 += h * h;
      }
    }
    sh_norm[t] = sqrt(s2);
  }
  __syncthreads();
  // ordered compaction of the occupied bins (nbins <= a few thousand: one pass per kLdThreads bins)
  __shared__ int wave_cnt[kLdThreads / 64];
  __shared__ int base_sh;
  if (t == 0) base_sh = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nbins; b0 += kLdThreads) {
    const int i = b0 + t;
    const unsigned long long k = i < nbins ? J.bins[i] : 0ull;
    const bool occ = i < nbins && k >= thres;
    const unsigned long long m = __ballot(occ);
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = base_sh;
    for (int w = 0; w < wave; w++) base += wave_cnt[w];
    if (occ) {
      const int o = base + __popcll(m & ((1ull << lane) - 1ull));
      J.sig_idx[o] = i;
      J.sig_val[o] = key_to_double(k) / sh_norm[i / num_r]; // :139-141
    }
    __syncthreads();
    if (t == 0) {
      int tot = 0;
      for (int w = 0; w < kLdThreads / 64; w++) tot += wave_cnt[w];
      base_sh += tot;
    }
    __syncthreads();
  }
  if (t == 0) *J.n_sig = base_sh;
}

template <class T>
int dev_alloc(std::vector<void *> &owned, T **p, size_t n) {
  *p = nullptr;
  DSM_HIP(hipMalloc((void **)p, sizeof(T) * (n ? n : 1)));
  owned.push_back(*p);
  return DSM_OK;
}

} // namespace

// ScanContext::generate, between the halves (ScanContext.cpp:41-64): 3 x 3 eigen-decomposition of the covariance (the host
// form's cyclic Jacobi, same operations in the same order: loopdet_internal.hpp), centroid, eigenvectors and tfm_pca_rig
// of every job -- one thread per job, on the device since round 3 (the moments used to travel to the host and back).
__global__ void sc_pca_kernel(JobDev *jobs, int n_jobs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  JobDev &J = jobs[j];
  if (!J.ringkey || *J.n_out < 1) return;
  const double *m = J.moments;
  const double cov[9] = {m[3], m[4], m[5], m[4], m[6], m[7], m[5], m[7], m[8]};
  double ev[3], V[9];
  dsm::eig3_sym(cov, ev, V);
  for (int i = 0; i < 3; i++) J.mean[i] = m[i];
  for (int i = 0; i < 9; i++) J.V[i] = V[i];
  double *tfm = J.tfm;
  for (int i = 0; i < 16; i++) tfm[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) tfm[r * 4 + c] = V[c * 3 + r];
  for (int r = 0; r < 3; r++) tfm[r * 4 + 3] = -(tfm[r * 4 + 0] * m[0] + tfm[r * 4 + 1] * m[1] + tfm[r * 4 + 2] * m[2]);
}

extern "C" {

int dsm_loop_descriptors_batch(dsm_context *ctx, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r) {
  if (!ctx || n_jobs < 1 || !jobs || !(lidar_range > 0) || num_s < 1 || num_r < 1 || num_s > kLdThreads || num_r > kLdThreads)
    return invalid("dsm_loop_descriptors_batch: bad argument");
  const long long vs0 = (long long)std::floor(2 * lidar_range * 1.0) + 1, vs1 = (long long)std::floor(2 * lidar_range * 2.0) + 1,
                  vs2 = (long long)std::floor(2 * lidar_range * 1.0) + 1; // :44-48
  const long long cells = vs0 * vs1 * vs2;
  if (cells > (1ll << 26)) return invalid("dsm_loop_descriptors_batch: lidar_range too large for the dense voxel grid (use the host form)");
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    if (J.n_pts < 0 || J.n_kf < 0 || !J.cur_cw || !J.n_out || (J.n_kf && (!J.kf_ids || !J.kf_pose_wc || !J.kf_keep)) ||
        (J.n_pts && (!J.pt_kf_id || !J.pt_xyz || !J.sel_idx || !J.pts_spherical)) ||
        (J.ringkey && (!J.sig_idx || !J.sig_val || !J.n_sig || !J.tfm_pca_rig)))
      return invalid("dsm_loop_descriptors_batch: bad job");
  }
  DSM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  std::vector<void *> owned;
  struct Free {
    std::vector<void *> &v;
    ~Free() {
      for (void *p : v) hipFree(p);
    }
  } guard{owned};
  const int nbins = num_s * num_r, nblocks = (int)((cells + kLdThreads - 1) / kLdThreads);
  std::vector<JobDev> hj(n_jobs);
  std::vector<std::vector<unsigned char>> keep(n_jobs);
  int max_pts = 1;
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    JobDev &D = hj[j];
    memset(&D, 0, sizeof D);
    D.n_pts = J.n_pts;
    if (J.n_pts > max_pts) max_pts = J.n_pts;
    memcpy(D.cw, J.cur_cw, sizeof D.cw);
    // :33-41 on the host (a handful of keyframes): which keyframes survive, hence which points
    trim_keyframes(J.n_kf, J.kf_pose_wc, J.cur_cw, J.kf_keep);
    std::vector<std::pair<int, int>> ids;
    for (int k = 0; k < J.n_kf; k++) ids.push_back(std::make_pair(J.kf_ids[k], J.kf_keep[k]));
    std::sort(ids.begin(), ids.end());
    keep[j].resize(J.n_pts ? J.n_pts : 1);
    for (int i = 0; i < J.n_pts; i++) {
      auto it = std::lower_bound(ids.begin(), ids.end(), std::make_pair(J.pt_kf_id[i], 0));
      unsigned char k = 0;
      for (; it != ids.end() && it->first == J.pt_kf_id[i]; ++it)
        if (it->second) k = 1;
      keep[j][i] = k; // unknown keyframe: `find == end` (:55)
    }
    double *xyz;
    unsigned char *kp;
    int rc;
    if ((rc = dev_alloc(owned, &xyz, 3 * (size_t)J.n_pts))) return rc;
    if ((rc = dev_alloc(owned, &kp, (size_t)J.n_pts))) return rc;
    if (J.n_pts) {
      DSM_HIP(hipMemcpyAsync(xyz, J.pt_xyz, sizeof(double) * 3 * (size_t)J.n_pts, hipMemcpyHostToDevice, st));
      DSM_HIP(hipMemcpyAsync(kp, keep[j].data(), (size_t)J.n_pts, hipMemcpyHostToDevice, st));
    }
    D.xyz = xyz, D.keep = kp;
    if ((rc = dev_alloc(owned, &D.grid_y, (size_t)cells))) return rc;
    if ((rc = dev_alloc(owned, &D.grid_idx, (size_t)cells))) return rc;
    if ((rc = dev_alloc(owned, &D.block_count, (size_t)nblocks))) return rc;
    if ((rc = dev_alloc(owned, &D.sel_idx, (size_t)J.n_pts))) return rc;
    if ((rc = dev_alloc(owned, &D.sph, 3 * (size_t)J.n_pts))) return rc;
    if ((rc = dev_alloc(owned, &D.n_out, 2))) return rc;
    D.n_sig = D.n_out + 1;
    if ((rc = dev_alloc(owned, &D.moments, 9))) return rc;
    if ((rc = dev_alloc(owned, &D.bins, (size_t)nbins))) return rc;
    if ((rc = dev_alloc(owned, &D.ringkey, (size_t)num_r))) return rc;
    if ((rc = dev_alloc(owned, &D.sig_idx, (size_t)nbins))) return rc;
    if ((rc = dev_alloc(owned, &D.sig_val, (size_t)nbins))) return rc;
  }
  JobDev *dj;
  int rc = dev_alloc(owned, &dj, (size_t)n_jobs);
  if (rc) return rc;
  DSM_HIP(hipMemcpyAsync(dj, hj.data(), sizeof(JobDev) * n_jobs, hipMemcpyHostToDevice, st));

  // ---- generate_spherical_points: all jobs side by side (blockIdx.y = job)
  const int gx_cells = (int)std::min<long long>(nblocks, 4096), gx_pts = std::min((max_pts + kLdThreads - 1) / kLdThreads, 1024);
  hipLaunchKernelGGL(voxel_clear_kernel, dim3(gx_cells, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  hipLaunchKernelGGL(voxel_min_y_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, vs0, vs1);
  hipLaunchKernelGGL(voxel_min_idx_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, vs0, vs1);
  hipLaunchKernelGGL(voxel_count_kernel, dim3(nblocks, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  hipLaunchKernelGGL(voxel_scan_kernel, dim3(n_jobs), dim3(kLdThreads), 0, st, dj, nblocks);
  hipLaunchKernelGGL(voxel_emit_kernel, dim3(nblocks, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  // ---- ScanContext::generate, first half: PCA moments
  bool any_sc = false;
  for (int j = 0; j < n_jobs; j++) any_sc = any_sc || jobs[j].ringkey;
  if (any_sc) {
    hipLaunchKernelGGL(sc_moments_kernel, dim3(n_jobs), dim3(64), 0, st, dj);
    // ---- the eigen-decomposition (:41-47) and tfm_pca_rig (:55-64), then the second half (binning, ring key, signature):
    // nothing leaves the device between the halves (jobs without points are skipped by every kernel and reported below)
    hipLaunchKernelGGL(sc_pca_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, st, dj, n_jobs);
    hipLaunchKernelGGL(sc_clear_kernel, dim3((nbins + kLdThreads - 1) / kLdThreads, n_jobs), dim3(kLdThreads), 0, st, dj, nbins, lidar_range);
    hipLaunchKernelGGL(sc_bin_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, num_s, num_r);
    hipLaunchKernelGGL(sc_finish_kernel, dim3(n_jobs), dim3(kLdThreads), sizeof(double) * num_s, st, dj, lidar_range, num_s, num_r);
  }
  DSM_HIP(hipGetLastError());
  std::vector<int> n_out(n_jobs);
  std::vector<JobDev> back(any_sc ? n_jobs : 0);
  for (int j = 0; j < n_jobs; j++) DSM_HIP(hipMemcpyAsync(&n_out[j], hj[j].n_out, sizeof(int), hipMemcpyDeviceToHost, st));
  if (any_sc) DSM_HIP(hipMemcpyAsync(back.data(), dj, sizeof(JobDev) * n_jobs, hipMemcpyDeviceToHost, st)); // tfm of every job
  DSM_HIP(hipStreamSynchronize(st));
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    *J.n_out = n_out[j];
    if (n_out[j] > 0) {
      DSM_HIP(hipMemcpyAsync(J.sel_idx, hj[j].sel_idx, sizeof(int) * n_out[j], hipMemcpyDeviceToHost, st));
      DSM_HIP(hipMemcpyAsync(J.pts_spherical, hj[j].sph, sizeof(double) * 3 * (size_t)n_out[j], hipMemcpyDeviceToHost, st));
    }
  }
  if (!any_sc) {
    DSM_HIP(hipStreamSynchronize(st));
    return DSM_OK;
  }
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    if (!J.ringkey) continue;
    if (n_out[j] < 1) {
      DSM_HIP(hipStreamSynchronize(st));
      return invalid("dsm_loop_descriptors_batch: ScanContext of an empty point set");
    }
    memcpy(J.tfm_pca_rig, back[j].tfm, sizeof(double) * 16);
  }
  std::vector<int> n_sig(n_jobs);
  for (int j = 0; j < n_jobs; j++) DSM_HIP(hipMemcpyAsync(&n_sig[j], hj[j].n_sig, sizeof(int), hipMemcpyDeviceToHost, st));
  DSM_HIP(hipStreamSynchronize(st));
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    if (!J.ringkey) continue;
    *J.n_sig = n_sig[j];
    DSM_HIP(hipMemcpyAsync(J.ringkey, hj[j].ringkey, sizeof(float) * num_r, hipMemcpyDeviceToHost, st));
    if (n_sig[j] > 0) {
      DSM_HIP(hipMemcpyAsync(J.sig_idx, hj[j].sig_idx, sizeof(int) * n_sig[j], hipMemcpyDeviceToHost, st));
      DSM_HIP(hipMemcpyAsync(J.sig_val, hj[j].sig_val, sizeof(double) * n_sig[j], hipMemcpyDeviceToHost, st));
    }
  }
  DSM_HIP(hipStreamSynchronize(st));
  return DSM_OK;
}

} // extern "C"
