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
#include "ringdb_internal.hpp"

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
  const int *pt_kf;         // n_pts: id of the keyframe that owns the point
  const int *ids;           // n_ids pairs (keyframe id, survived the orientation trim :33-41), sorted
  int n_ids, n_pts;
  // the caller's cloud in page-locked memory (device-visible addresses): loop_gather_kernel copies it into xyz / pt_kf -- no staging copy
  // on the host; null: the cloud travelled through the pinned mirror
  const double *xyz_src;
  const int *pt_kf_src;
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
  double *tfm_out;            // ... and tfm_pca_rig (:55-64), 16 doubles in the output region
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
  { // :55: a point whose keyframe is unknown or was trimmed is dropped (an id listed twice is kept if either entry survived). A few
    // keyframes per window: the search over the sorted pairs costs less here than the per-point host loop it replaces (0.24 ms per
    // 16 000-point keyframe: most of the fused chain's time at 64 keyframes per call)
    const int id = J.pt_kf[i];
    int lo = 0, hi = J.n_ids;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (J.ids[2 * mid] < id) lo = mid + 1; else hi = mid;
    }
    bool kept = false;
    for (; lo < J.n_ids && J.ids[2 * lo] == id; lo++) kept = kept || J.ids[2 * lo + 1] != 0;
    if (!kept) return -1;
  }
  to_camera(J.cw, J.xyz + 3 * (size_t)i, p);
  if (sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) >= lidar_range) return -1;
  const long long xi = (long long)floor((p[0] + lidar_range) * 1.0), yi = (long long)floor((p[1] + lidar_range) * 2.0),
                  zi = (long long)floor((p[2] + lidar_range) * 1.0); // steps 1/RES_X, 1/RES_Y, 1/RES_Z (:23-25,:44)
  return xi + yi * vs0 + zi * vs0 * vs1;
}

// the clouds of jobs whose caller keeps them in page-locked memory: read over the bus by the kernel, 8 bytes per access
__global__ void loop_gather_kernel(const JobDev *jobs) {
  const JobDev &J = jobs[blockIdx.y];
  if (!J.xyz_src) return;
  const size_t nd = 3 * (size_t)J.n_pts, stride = (size_t)gridDim.x * blockDim.x;
  double *xyz = const_cast<double *>(J.xyz);
  int *pk = const_cast<int *>(J.pt_kf);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += stride) xyz[i] = __builtin_nontemporal_load(J.xyz_src + i);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)J.n_pts; i += stride) pk[i] = __builtin_nontemporal_load(J.pt_kf_src + i);
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
// align_points_PCA :22-40.  The mean is the reference's loop to the letter -- three sums in point order (lanes 0 .. 2 of wave 0) -- with
// the points travelling through LDS tiles that waves 1 .. 3 load one tile ahead (coalesced), so that the chain is the additions alone
// and not a global-memory round trip per point; the covariance as kCovLanes interleaved partial sums added in ascending order
// (loopdet_internal.hpp), one partial per thread: the host form's order and bits.
constexpr int kMomTile = 1024; // points per LDS tile
static_assert(kCovLanes == kLdThreads, "one covariance partial per thread");
__global__ __launch_bounds__(kLdThreads) void sc_moments_kernel(const JobDev *jobs) {
  const JobDev &J = jobs[blockIdx.x];
  const int n = *J.n_out, t = threadIdx.x;
  __shared__ double tile[2][3 * kMomTile];
  __shared__ double mean[3];
  __shared__ double part[6][kCovLanes];
  const int ntiles = (n + kMomTile - 1) / kMomTile;
  auto load_tile = [&](int k, int first, int step) {
    const size_t base = 3 * (size_t)k * kMomTile;
    const int cnt = 3 * ((n - k * kMomTile) < kMomTile ? (n - k * kMomTile) : kMomTile);
    for (int i = first; i < cnt; i += step) tile[k & 1][i] = J.sph[base + i];
  };
  if (ntiles) load_tile(0, t, kLdThreads);
  __syncthreads();
  double s = 0;
  for (int k = 0; k < ntiles; k++) {
    if (t >= 64) {
      if (k + 1 < ntiles) load_tile(k + 1, t - 64, kLdThreads - 64);
    } else if (t < 3) {
      const int cnt = (n - k * kMomTile) < kMomTile ? (n - k * kMomTile) : kMomTile;
      const double *p = &tile[k & 1][t];
      for (int i = 0; i < cnt; i++) s += p[3 * i];
    }
    __syncthreads();
  }
  if (t < 3) {
    mean[t] = n > 0 ? s / n : 0.0;
    J.moments[t] = mean[t];
  }
  __syncthreads();
  {
    const double mx = mean[0], my = mean[1], mz = mean[2];
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (int i = t; i < n; i += kCovLanes) {
      const double x = J.sph[3 * (size_t)i] - mx, y = J.sph[3 * (size_t)i + 1] - my, z = J.sph[3 * (size_t)i + 2] - mz;
      xx += x * x, xy += x * y, xz += x * z, yy += y * y, yz += y * z, zz += z * z;
    }
    part[0][t] = xx, part[1][t] = xy, part[2][t] = xz, part[3][t] = yy, part[4][t] = yz, part[5][t] = zz;
  }
  __syncthreads();
  if (t < 6) {
    double acc = part[t][0];
    for (int l = 1; l < kCovLanes; l++) acc += part[t][l];
    J.moments[3 + t] = acc; // xx xy xz yy yz zz
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
  double *tfm = J.tfm_out;
  for (int i = 0; i < 16; i++) tfm[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) tfm[r * 4 + c] = V[c * 3 + r];
  for (int r = 0; r < 3; r++) tfm[r * 4 + 3] = -(tfm[r * 4 + 0] * m[0] + tfm[r * 4 + 1] * m[1] + tfm[r * 4 + 2] * m[2]);
}

// ---- host side: ONE enqueue and ONE read-back per call -------------------------------------------------------------------------------
// Round 4's form allocated thirteen device buffers per job and freed them at the end (hipFree synchronises), staged its inputs from
// pageable memory job by job, synchronised three times and copied every small output back on its own: 0.9 ms of host time for ONE
// keyframe around ~60 us of kernels.  Now: a device arena and a page-locked mirror that live in the context (grown on demand), the
// inputs of all jobs packed into one host->device copy, the outputs of all jobs packed into one device->host copy, and -- for
// dsm_loop_detect_batch -- the ring keys handed to the ring-key search ON THE DEVICE between the two.
namespace {

struct Arena { // linear sub-allocation of a region that is laid out identically on the device and in the pinned mirror
  size_t used = 0;
  size_t take(size_t bytes) {
    const size_t o = used;
    used = (used + bytes + 255) & ~(size_t)255;
    return o;
  }
};

struct LoopPlan {
  int n_jobs = 0, nbins = 0, num_r = 0, nblocks = 0;
  long long cells = 0;
  std::vector<JobDev> hj;
  // offsets (bytes) into the input region (pinned + device), the work region (device only), the output region (device + pinned)
  std::vector<size_t> in_xyz, in_ptkf, in_ids, out_small, out_sig_idx, out_sig_val, out_sel, out_sph;
  size_t in_tab = 0, in_bytes = 0, work_bytes = 0, out_bytes = 0, out_keys = 0, out_cand = 0;
  bool any_sc = false;
};
// per job "small" outputs: [n_out, n_sig, pad, pad][tfm 16 doubles] -- ring keys of all jobs are one contiguous array (out_keys)
constexpr size_t kSmallBytes = 16 + 16 * sizeof(double);

int grow(dsm_context *ctx, size_t dev_bytes, size_t pin_bytes) {
  if (dev_bytes > ctx->loop_dev_bytes) {
    if (ctx->loop_dev) DSM_HIP(hipFree(ctx->loop_dev));
    ctx->loop_dev = nullptr, ctx->loop_dev_bytes = 0;
    DSM_HIP(hipMalloc(&ctx->loop_dev, dev_bytes + dev_bytes / 2));
    ctx->loop_dev_bytes = dev_bytes + dev_bytes / 2;
  }
  if (pin_bytes > ctx->loop_pin_bytes) {
    if (ctx->loop_pin) DSM_HIP(hipHostFree(ctx->loop_pin));
    ctx->loop_pin = nullptr, ctx->loop_pin_bytes = 0;
    DSM_HIP(hipHostMalloc(&ctx->loop_pin, pin_bytes + pin_bytes / 2, hipHostMallocDefault));
    ctx->loop_pin_bytes = pin_bytes + pin_bytes / 2;
  }
  return DSM_OK;
}

int check_jobs(dsm_context *ctx, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r, long long *cells_out) {
  if (!ctx || n_jobs < 1 || !jobs || !(lidar_range > 0) || num_s < 1 || num_r < 1 || num_s > kLdThreads || num_r > kLdThreads)
    return invalid("dsm_loop_descriptors_batch: bad argument");
  const long long vs0 = (long long)std::floor(2 * lidar_range * 1.0) + 1, vs1 = (long long)std::floor(2 * lidar_range * 2.0) + 1,
                  vs2 = (long long)std::floor(2 * lidar_range * 1.0) + 1; // :44-48
  const long long cells = vs0 * vs1 * vs2;
  if (cells > (1ll << 26)) return invalid("dsm_loop_descriptors_batch: lidar_range too large for the dense voxel grid (use the host form)");
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    if (J.n_pts < 0 || J.n_kf < 0 || !J.cur_cw || !J.n_out || (J.n_kf && (!J.kf_ids || !J.kf_pose_wc || !J.kf_keep)) || (J.n_pts && (!J.pt_kf_id || !J.pt_xyz)) ||
        ((J.sel_idx == nullptr) != (J.pts_spherical == nullptr)) || (J.ringkey && (!J.sig_idx || !J.sig_val || !J.n_sig || !J.tfm_pca_rig)))
      return invalid("dsm_loop_descriptors_batch: bad job");
  }
  *cells_out = cells;
  return DSM_OK;
}

// everything up to (and without) the read-back: inputs staged and copied, kernels enqueued.  extra_out_bytes: room at the end of the
// output region for the caller's own device results (the ring-key search's packed candidates); *d_keys / *d_extra: where they live
int loop_enqueue(dsm_context *ctx, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r, size_t extra_out_bytes, LoopPlan &P,
                 float **d_keys, void **d_extra) {
  long long cells = 0;
  int rc = check_jobs(ctx, n_jobs, jobs, lidar_range, num_s, num_r, &cells);
  if (rc) return rc;
  const long long vs0 = (long long)std::floor(2 * lidar_range * 1.0) + 1, vs1 = (long long)std::floor(2 * lidar_range * 2.0) + 1;
  DSM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  P.n_jobs = n_jobs, P.nbins = num_s * num_r, P.num_r = num_r, P.cells = cells, P.nblocks = (int)((cells + kLdThreads - 1) / kLdThreads);
  P.hj.assign(n_jobs, JobDev());
  P.in_xyz.resize(n_jobs), P.in_ptkf.resize(n_jobs), P.in_ids.resize(n_jobs), P.out_small.resize(n_jobs), P.out_sig_idx.resize(n_jobs), P.out_sig_val.resize(n_jobs);
  P.out_sel.resize(n_jobs), P.out_sph.resize(n_jobs);
  Arena in, work, out;
  std::vector<size_t> w_gy(n_jobs), w_gi(n_jobs), w_bc(n_jobs), w_mom(n_jobs), w_bins(n_jobs), w_sel(n_jobs), w_sph(n_jobs);
  int max_pts = 1;
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    if (J.n_pts > max_pts) max_pts = J.n_pts;
    P.any_sc = P.any_sc || J.ringkey;
    P.in_ids[j] = in.take(sizeof(int) * 2 * (size_t)(J.n_kf > 0 ? J.n_kf : 1));
    w_gy[j] = work.take(sizeof(unsigned long long) * (size_t)cells);
    w_gi[j] = work.take(sizeof(int) * (size_t)cells);
    w_bc[j] = work.take(sizeof(int) * (size_t)P.nblocks);
    w_mom[j] = work.take(sizeof(double) * 9);
    w_bins[j] = work.take(sizeof(unsigned long long) * (size_t)P.nbins);
    P.out_small[j] = out.take(kSmallBytes);
    P.out_sig_idx[j] = out.take(sizeof(int) * (size_t)P.nbins);
    P.out_sig_val[j] = out.take(sizeof(double) * (size_t)P.nbins);
    // the selected points: read back only where the caller asks for them (0.45 MB per keyframe)
    if (J.sel_idx) {
      P.out_sel[j] = out.take(sizeof(int) * (size_t)J.n_pts);
      P.out_sph[j] = out.take(sizeof(double) * 3 * (size_t)J.n_pts);
    } else {
      w_sel[j] = work.take(sizeof(int) * (size_t)J.n_pts);
      w_sph[j] = work.take(sizeof(double) * 3 * (size_t)J.n_pts);
    }
  }
  P.in_tab = in.take(sizeof(JobDev) * (size_t)n_jobs);
  const size_t in_small_bytes = in.used; // [keyframe tables | job table], then the clouds: clouds read directly from page-locked caller
                                         // memory (loop_gather_kernel) do not travel through the mirror at all
  for (int j = 0; j < n_jobs; j++) {
    P.in_xyz[j] = in.take(sizeof(double) * 3 * (size_t)jobs[j].n_pts);
    P.in_ptkf[j] = in.take(sizeof(int) * (size_t)jobs[j].n_pts);
  }
  P.out_keys = out.take(sizeof(float) * (size_t)num_r * n_jobs);
  P.out_cand = out.take(extra_out_bytes);
  P.in_bytes = in.used, P.work_bytes = work.used, P.out_bytes = out.used;
  // device arena: [in | work | out]; pinned mirror: [in | out]
  rc = grow(ctx, P.in_bytes + P.work_bytes + P.out_bytes, P.in_bytes + P.out_bytes);
  if (rc) return rc;
  unsigned char *d_in = (unsigned char *)ctx->loop_dev, *d_work = d_in + P.in_bytes, *d_out = d_work + P.work_bytes;
  unsigned char *h_in = (unsigned char *)ctx->loop_pin;
  bool any_direct = false, any_staged = false;
  for (int j = 0; j < n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    JobDev &D = P.hj[j];
    memset(&D, 0, sizeof D);
    D.n_pts = J.n_pts;
    memcpy(D.cw, J.cur_cw, sizeof D.cw);
    // :33-41 on the host (a handful of keyframes): which keyframes survive, hence which points
    trim_keyframes(J.n_kf, J.kf_pose_wc, J.cur_cw, J.kf_keep);
    std::vector<std::pair<int, int>> ids;
    for (int k = 0; k < J.n_kf; k++) ids.push_back(std::make_pair(J.kf_ids[k], J.kf_keep[k] ? 1 : 0));
    std::sort(ids.begin(), ids.end());
    int *h_ids = (int *)(h_in + P.in_ids[j]);
    for (int k = 0; k < J.n_kf; k++) h_ids[2 * k] = ids[k].first, h_ids[2 * k + 1] = ids[k].second;
    // the cloud: straight from the caller's buffers where both are page-locked (dsm_host_alloc / hipHostMalloc / hipHostRegister: a node
    // that keeps its window's clouds there), else through the pinned mirror (a host copy of 28 bytes per point)
    const void *dev_xyz = nullptr, *dev_kf = nullptr;
    if (J.n_pts) {
      hipPointerAttribute_t ax, ak;
      if (hipPointerGetAttributes(&ax, J.pt_xyz) == hipSuccess && ax.type == hipMemoryTypeHost && ax.devicePointer &&
          hipPointerGetAttributes(&ak, J.pt_kf_id) == hipSuccess && ak.type == hipMemoryTypeHost && ak.devicePointer)
        dev_xyz = ax.devicePointer, dev_kf = ak.devicePointer;
      else
        (void)hipGetLastError(); // (pageable memory is reported as an error by some runtimes)
    }
    if (dev_xyz) {
      D.xyz_src = (const double *)dev_xyz, D.pt_kf_src = (const int *)dev_kf;
      any_direct = true;
    } else if (J.n_pts) {
      memcpy(h_in + P.in_ptkf[j], J.pt_kf_id, sizeof(int) * (size_t)J.n_pts);
      memcpy(h_in + P.in_xyz[j], J.pt_xyz, sizeof(double) * 3 * (size_t)J.n_pts);
      any_staged = true;
    }
    D.xyz = (const double *)(d_in + P.in_xyz[j]), D.pt_kf = (const int *)(d_in + P.in_ptkf[j]);
    D.ids = (const int *)(d_in + P.in_ids[j]), D.n_ids = J.n_kf;
    D.grid_y = (unsigned long long *)(d_work + w_gy[j]), D.grid_idx = (int *)(d_work + w_gi[j]), D.block_count = (int *)(d_work + w_bc[j]);
    D.moments = (double *)(d_work + w_mom[j]), D.bins = (unsigned long long *)(d_work + w_bins[j]);
    D.sel_idx = (int *)(J.sel_idx ? d_out + P.out_sel[j] : d_work + w_sel[j]);
    D.sph = (double *)(J.sel_idx ? d_out + P.out_sph[j] : d_work + w_sph[j]);
    D.n_out = (int *)(d_out + P.out_small[j]);
    D.n_sig = D.n_out + 1;
    D.tfm_out = (double *)(d_out + P.out_small[j] + 16);
    D.ringkey = J.ringkey ? (float *)(d_out + P.out_keys) + (size_t)j * num_r : nullptr;
    D.sig_idx = (int *)(d_out + P.out_sig_idx[j]), D.sig_val = (double *)(d_out + P.out_sig_val[j]);
  }
  memcpy(h_in + P.in_tab, P.hj.data(), sizeof(JobDev) * (size_t)n_jobs);
  DSM_HIP(hipMemcpyAsync(d_in, h_in, any_staged ? P.in_bytes : in_small_bytes, hipMemcpyHostToDevice, st));
  JobDev *dj = (JobDev *)(d_in + P.in_tab);
  if (any_direct) hipLaunchKernelGGL(loop_gather_kernel, dim3(64, n_jobs), dim3(kLdThreads), 0, st, dj);
  // a job without a descriptor leaves its key slot untouched: zero the key array so that the search sees defined values
  DSM_HIP(hipMemsetAsync(d_out + P.out_keys, 0, sizeof(float) * (size_t)num_r * n_jobs, st));
  // ---- generate_spherical_points: all jobs side by side (blockIdx.y = job)
  const int gx_cells = (int)std::min<long long>(P.nblocks, 4096), gx_pts = std::min((max_pts + kLdThreads - 1) / kLdThreads, 1024);
  hipLaunchKernelGGL(voxel_clear_kernel, dim3(gx_cells, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  hipLaunchKernelGGL(voxel_min_y_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, vs0, vs1);
  hipLaunchKernelGGL(voxel_min_idx_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, vs0, vs1);
  hipLaunchKernelGGL(voxel_count_kernel, dim3(P.nblocks, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  hipLaunchKernelGGL(voxel_scan_kernel, dim3(n_jobs), dim3(kLdThreads), 0, st, dj, P.nblocks);
  hipLaunchKernelGGL(voxel_emit_kernel, dim3(P.nblocks, n_jobs), dim3(kLdThreads), 0, st, dj, cells);
  if (P.any_sc) {
    // ---- ScanContext::generate: PCA moments, the eigen-decomposition (:41-47) and tfm_pca_rig (:55-64), then binning, ring key and
    // signature: nothing leaves the device between the halves (jobs without points are skipped by every kernel and reported at the end)
    hipLaunchKernelGGL(sc_moments_kernel, dim3(n_jobs), dim3(kLdThreads), 0, st, dj);
    hipLaunchKernelGGL(sc_pca_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, st, dj, n_jobs);
    hipLaunchKernelGGL(sc_clear_kernel, dim3((P.nbins + kLdThreads - 1) / kLdThreads, n_jobs), dim3(kLdThreads), 0, st, dj, P.nbins, lidar_range);
    hipLaunchKernelGGL(sc_bin_kernel, dim3(gx_pts, n_jobs), dim3(kLdThreads), 0, st, dj, lidar_range, num_s, num_r);
    hipLaunchKernelGGL(sc_finish_kernel, dim3(n_jobs), dim3(kLdThreads), sizeof(double) * num_s, st, dj, lidar_range, num_s, num_r);
  }
  DSM_HIP(hipGetLastError());
  if (d_keys) *d_keys = (float *)(d_out + P.out_keys);
  if (d_extra) *d_extra = d_out + P.out_cand;
  return DSM_OK;
}

// the ONE read-back and the scatter into the caller's arrays; h_extra: where the caller's own results arrived
int loop_finish(dsm_context *ctx, const dsm_loop_job *jobs, const LoopPlan &P, const void **h_extra) {
  hipStream_t st = ctx->stream;
  unsigned char *d_out = (unsigned char *)ctx->loop_dev + P.in_bytes + P.work_bytes;
  unsigned char *h_out = (unsigned char *)ctx->loop_pin + P.in_bytes;
  DSM_HIP(hipMemcpyAsync(h_out, d_out, P.out_bytes, hipMemcpyDeviceToHost, st));
  DSM_HIP(hipStreamSynchronize(st));
  if (h_extra) *h_extra = h_out + P.out_cand;
  // A job that asks for a descriptor and whose cloud came out empty fails the WHOLE call before anything is written to the caller's
  // arrays (and, in dsm_loop_detect_batch, before any key is enqueued): a batch is all or nothing, unlike the sequential calls it
  // otherwise equals, which would have failed at that keyframe with the earlier ones already through
  for (int j = 0; j < P.n_jobs; j++)
    if (jobs[j].ringkey && ((const int *)(h_out + P.out_small[j]))[0] < 1)
      return invalid("dsm_loop_descriptors_batch: ScanContext of an empty point set (no output of the call was written)");
  for (int j = 0; j < P.n_jobs; j++) {
    const dsm_loop_job &J = jobs[j];
    const int *small = (const int *)(h_out + P.out_small[j]);
    const int n_out = small[0], n_sig = small[1];
    *J.n_out = n_out;
    if (J.sel_idx && n_out > 0) {
      memcpy(J.sel_idx, h_out + P.out_sel[j], sizeof(int) * (size_t)n_out);
      memcpy(J.pts_spherical, h_out + P.out_sph[j], sizeof(double) * 3 * (size_t)n_out);
    }
    if (!J.ringkey) continue;
    memcpy(J.tfm_pca_rig, h_out + P.out_small[j] + 16, sizeof(double) * 16);
    *J.n_sig = n_sig;
    memcpy(J.ringkey, (const float *)(h_out + P.out_keys) + (size_t)j * P.num_r, sizeof(float) * (size_t)P.num_r);
    if (n_sig > 0) {
      memcpy(J.sig_idx, h_out + P.out_sig_idx[j], sizeof(int) * (size_t)n_sig);
      memcpy(J.sig_val, h_out + P.out_sig_val[j], sizeof(double) * (size_t)n_sig);
    }
  }
  return DSM_OK;
}

} // namespace

extern "C" {

int dsm_loop_descriptors_batch(dsm_context *ctx, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r) {
  LoopPlan P;
  int rc = loop_enqueue(ctx, n_jobs, jobs, lidar_range, num_s, num_r, 0, P, nullptr, nullptr);
  if (rc) return rc;
  return loop_finish(ctx, jobs, P, nullptr);
}

// The per-keyframe loop chain of LoopHandler (LoopHandler.cpp:186 generate_spherical_points, :236 ScanContext::generate, :247
// search_ringkey) as ONE enqueue and ONE read-back, batched over the keyframes that were marginalised in the same advance (one per
// concurrent sequence): descriptors on the device, their ring keys handed to the ring-key index's k-NN on the device, then a single copy
// back of descriptors and candidates.  Semantics = the jobs' search_ringkey calls one after the other (search_place.h:25-57): query j
// searches the index as it stands after the keys of queries 0 .. j-1 were enqueued.  An enqueue moves the key that has waited for
// `margin` insertions from the delay queue into the index; those keys are OLDER than this batch (n_jobs <= margin), known to the host,
// few, and visible to the later queries of the batch only: the device searches the index as it stood before the batch, the host adds
// every query's distances to the keys that matured before it (FLANN's L2 expression, the kernels' own) and merges -- same candidates
// as the sequential calls, bit for bit.  cand_out: n_jobs x k indices (search_ringkey's `idx - 1`), ncand_out: n_jobs counts.
int dsm_loop_detect_batch(dsm_context *ctx, dsm_ringdb *db, int n_jobs, const dsm_loop_job *jobs, double lidar_range, int num_s, int num_r,
                          int *cand_out, int *ncand_out) {
  if (!db || !cand_out || !ncand_out) return invalid("dsm_loop_detect_batch: bad argument");
  if (db->ctx != ctx) return invalid("dsm_loop_detect_batch: the index belongs to another context");
  if (db->shard_count != 1) return invalid("dsm_loop_detect_batch: unsharded index only (a sharded one searches through dsm_ringdb_query_then_enqueue)");
  if (num_r != db->dim) return invalid("dsm_loop_detect_batch: num_r must equal the index's key dimension");
  if (n_jobs > db->margin) return invalid("dsm_loop_detect_batch: at most `margin` keyframes per call");
  for (int j = 0; jobs && j < n_jobs; j++)
    if (!jobs[j].ringkey) return invalid("dsm_loop_detect_batch: every job needs its descriptor outputs");
  LoopPlan P;
  float *d_keys = nullptr;
  void *d_cand = nullptr;
  const int k = db->k;
  int rc = loop_enqueue(ctx, n_jobs, jobs, lidar_range, num_s, num_r, sizeof(unsigned long long) * (size_t)n_jobs * k, P, &d_keys, &d_cand);
  if (rc) return rc;
  const bool search = db->size_global > k; // `ringkeys->size() > FLANN_NN`, search_place.h:29 (the index as it stands before the batch)
  if (search && (rc = dsm::ringdb_knn_device(db, d_keys, n_jobs, (unsigned long long *)d_cand))) return rc;
  const void *h_cand_v = nullptr;
  rc = loop_finish(ctx, jobs, P, &h_cand_v);
  if (rc) return rc;
  const unsigned long long *h_cand = (const unsigned long long *)h_cand_v;
  if (!search) {
    // The index is still too small to be searched (search_place.h:29) and may grow past that inside this batch: the first keyframes of a
    // run take the calls one by one (their descriptors are already here; exact by definition, and a handful of calls in a run's life)
    for (int j = 0; j < n_jobs; j++) {
      rc = dsm_ringdb_query_then_enqueue(db, jobs[j].ringkey, cand_out + (size_t)j * k, ncand_out + j);
      if (rc) return rc;
    }
    return DSM_OK;
  }
  // the sequential semantics on the host: matured keys and the queries that see them
  std::vector<float> matured; // keys that entered the index during this batch, in order
  const long long base = db->size_global;
  for (int j = 0; j < n_jobs; j++) {
    const float *key = jobs[j].ringkey;
    unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    const long long size_now = base + (long long)(matured.size() / db->dim);
    int nb = 0;
    if (size_now > k) {
      if (search)
        for (int i = 0; i < k; i++)
          if (h_cand[(size_t)j * k + i] != (unsigned long long)DSM_RINGDB_NO_CANDIDATE) best[nb++] = h_cand[(size_t)j * k + i];
      for (size_t m = 0; m < matured.size() / db->dim; m++) { // flann::L2, as the kernels: groups of four, then the tail
        const float *kp = matured.data() + m * db->dim;
        float result = 0.f;
        int d = 0;
        for (; d + 3 < db->dim; d += 4) {
          const float d0 = key[d] - kp[d], d1 = key[d + 1] - kp[d + 1], d2 = key[d + 2] - kp[d + 2], d3 = key[d + 3] - kp[d + 3];
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        for (; d < db->dim; d++) {
          const float d0 = key[d] - kp[d];
          result += d0 * d0;
        }
        if (!(result < db->thres)) continue;
        unsigned bits;
        memcpy(&bits, &result, 4);
        const unsigned long long c = ((unsigned long long)bits << 32) | (unsigned long long)(base + (long long)m);
        // insert into the ascending list of at most k
        int pos = nb < k ? nb : k;
        for (int i = 0; i < nb && i < k; i++)
          if (c < best[i]) {
            pos = i;
            break;
          }
        if (pos < k) {
          for (int i = (nb < k ? nb : k - 1); i > pos; i--) best[i] = best[i - 1];
          best[pos] = c;
          if (nb < k) nb++;
        }
      }
    }
    int nc = 0;
    for (int i = 0; i < nb && i < k; i++) {
      const int idx = (int)(best[i] & 0xFFFFFFFFull);
      if (idx > 0) cand_out[(size_t)j * k + nc++] = idx - 1; // :34-38
    }
    ncand_out[j] = nc;
    // the enqueue of search_ringkey (:41-56): the slot's old key matures into the index
    float *slot = db->queue.data() + (size_t)(db->queue_idx % db->margin) * db->dim;
    if (db->queue_idx >= db->margin) matured.insert(matured.end(), slot, slot + db->dim);
    memcpy(slot, key, sizeof(float) * db->dim);
    db->queue_idx++;
  }
  if (!matured.empty()) return dsm_ringdb_add_points(db, matured.data(), (int64_t)(matured.size() / db->dim));
  return DSM_OK;
}

} // extern "C"
