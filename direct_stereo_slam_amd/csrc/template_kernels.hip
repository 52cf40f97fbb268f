// template_kernels.hip -- row A4 / N3 on the device: TrackerAndScaler::makeCoarseDepthL0
// (TrackerAndScaler.cpp:143-315) from flat arrays of the window's active points straight into the
// tracker's float4 template, without the per-keyframe host loops and the host->device upload of the
// template lists.  Every float operation is the reference's, in the reference's order:
//   * points that splat onto the same pixel are summed in point order (:149-164) -- per-pixel lists
//     are linked with an atomic exchange (order irrelevant) and then walked in ascending point index;
//   * 2x2 sums (:166-187) and the 4-neighbour dilation (:190-275) are element-wise; the dilation reads
//     the pre-dilation weights (the reference's _bak copy) and only pre-dilation depths of pixels it
//     never writes, so a two-buffer form is identical to the in-place loop;
//   * the emit loop (:278-314) is an ordered stream compaction over the interior pixels in row-major
//     order (block counts -> exclusive scan -> ordered writes), so the template order -- and with it
//     the chunking and every reduction of the eval kernels -- equals the reference's emit order.
// The idepth_/weight_sums_ side products (only read by debugPlotIDepthMap) are not kept.
#include "dsm_kernels.hpp"

namespace dsm {

// One job = one keyframe's template (dsm_set_refs_from_points builds the templates of every sequence that made a keyframe in the same
// advance with ONE launch sequence: blockIdx.y = job; a per-job launch sequence was 26 launches x the jobs of a call, 1.1 of the 1.9 ms a
// node serving 128 sequences spent between two advances).  All jobs of a call share w, h and the level count.
struct TplView { // the pieces of a job's workspace (TplJob::ws), as launch_make_coarse_depth lays them out
  const float *pu, *pv, *pid, *pw;
  int *next, *head;
  float *idA, *wsA, *idB, *wsB;
  int *block_count;
};
__device__ __forceinline__ TplView tpl_view(const TplJob &J, int w, int h, int px_total) {
  TplView V;
  V.pu = J.pt, V.pv = V.pu + J.npts, V.pid = V.pv + J.npts, V.pw = V.pid + J.npts;
  V.next = (int *)J.ws;
  V.head = V.next + J.npts;
  V.idA = (float *)(V.head + (size_t)w * h);
  V.wsA = V.idA + px_total, V.idB = V.wsA + px_total, V.wsB = V.idB + px_total;
  V.block_count = (int *)(V.wsB + px_total);
  return V;
}

__global__ void tpl_init_kernel(const TplJob *__restrict__ jobs, int w, int h, int nlevels, int px_total) {
  const TplJob J = jobs[blockIdx.y];
  const TplView V = tpl_view(J, w, h, px_total);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < w * h) V.head[p] = -1;
  if (p <= nlevels) J.d_n[p] = 0;
}

__global__ void tpl_splat_link_kernel(const TplJob *__restrict__ jobs, int w, int h, int nlevels, int px_total) {
  const TplJob J = jobs[blockIdx.y];
  const TplView V = tpl_view(J, w, h, px_total);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= J.npts) return;
  const int u = (int)(V.pu[k] + 0.5f); // :151-152
  const int v = (int)(V.pv[k] + 0.5f);
  if (u < 0 || v < 0 || u >= w || v >= h) { // the reference would write out of bounds
    atomicOr(J.d_n + nlevels, 1);
    V.next[k] = -2;
    return;
  }
  V.next[k] = atomicExch(&V.head[u + w * v], k);
}

// one thread per level-0 pixel: sum this pixel's points in ascending point index (:160-161)
__global__ void tpl_splat_sum_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total) {
  const TplJob J = jobs[blockIdx.y];
  const TplView V = tpl_view(J, w, h, px_total);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= w * h) return;
  float sid = 0.f, sw = 0.f;
  const int hd = V.head[p];
  int last = -1;
  while (hd >= 0) {
    int best = 0x7FFFFFFF;
    for (int j = hd; j >= 0; j = V.next[j])
      if (j > last && j < best) best = j;
    if (best == 0x7FFFFFFF) break;
    sid += V.pid[best] * V.pw[best];
    sw += V.pw[best];
    last = best;
  }
  V.idA[p] = sid;
  V.wsA[p] = sw;
}

__global__ void tpl_pyr_sum_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total, int wl, int hl, int wlm1, int prev, int off) {
  const TplView V = tpl_view(jobs[blockIdx.y], w, h, px_total);
  const float *idm = V.idA + prev, *wsm = V.wsA + prev;
  float *idl = V.idA + off, *wsl = V.wsA + off;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wl * hl) return;
  const int x = i % wl, y = i / wl;
  const int b = 2 * x + 2 * y * wlm1;
  idl[i] = idm[b] + idm[b + 1] + idm[b + wlm1] + idm[b + wlm1 + 1];
  wsl[i] = wsm[b] + wsm[b + 1] + wsm[b + wlm1] + wsm[b + wlm1 + 1];
}

// :190-233 (levels 0-1, diagonal neighbours) / :236-275 (levels >= 2, axis neighbours)
__global__ void tpl_dilate_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total, int wl, int hl, int diagonal, int off) {
  const TplView V = tpl_view(jobs[blockIdx.y], w, h, px_total);
  const float *idl_in = V.idA + off, *bak = V.wsA + off;
  float *idl_out = V.idB + off, *ws_out = V.wsB + off;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int npix = wl * hl;
  if (i >= npix) return;
  float id = idl_in[i], ws = bak[i];
  if (i >= wl && i < npix - wl && ws <= 0) {
    const int o0 = diagonal ? 1 + wl : 1, o1 = diagonal ? -1 - wl : -1, o2 = diagonal ? wl - 1 : wl, o3 = diagonal ? -wl + 1 : -wl;
    const int offs[4] = {o0, o1, o2, o3};
    float sum = 0, num = 0, numn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (bak[i + offs[k]] > 0) {
        sum += idl_in[i + offs[k]];
        num += bak[i + offs[k]];
        numn++;
      }
    if (numn > 0) {
      id = sum / numn;
      ws = num / numn;
    }
  }
  idl_out[i] = id;
  ws_out[i] = ws;
}

constexpr int kEmitThreads = 256, kEmitItems = 4, kEmitBlock = kEmitThreads * kEmitItems;

// the emit test of one interior item (:291-307); returns the template entry in `e`
__device__ __forceinline__ bool tpl_emit_item(int item, int wi, int wl, const float *idl, const float *ws, const float *ref,
                                              int texel_floats, float4 &e) {
  const int x = 2 + item % wi, y = 2 + item / wi;
  const int i = x + y * wl;
  const float wsum = ws[i];
  if (!(wsum > 0)) return false;
  const float id = idl[i] / wsum;
  const float color = ref[(size_t)texel_floats * i];
  e = make_float4((float)x, (float)y, id, color);
  return __builtin_isfinite(color) && id > 0;
}

__global__ __launch_bounds__(kEmitThreads) void tpl_emit_count_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total, int lvl,
                                                                      int nitems, int wi, int wl, int off, int texel_floats) {
  const TplJob J = jobs[blockIdx.y];
  const TplView V = tpl_view(J, w, h, px_total);
  const float *idl = V.idB + off, *ws = V.wsB + off, *ref = J.ref[lvl];
  __shared__ int wave_cnt[kEmitThreads / 64];
  const int base = blockIdx.x * kEmitBlock;
  int c = 0;
  for (int j = 0; j < kEmitItems; j++) {
    const int item = base + j * kEmitThreads + threadIdx.x;
    float4 e;
    const bool v = item < nitems && tpl_emit_item(item, wi, wl, idl, ws, ref, texel_floats, e);
    c += __builtin_popcountll(__ballot(v));
  }
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) V.block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of the block counts by one workgroup per job (<= 1024 counts per pass, carried), total -> d_n[lvl]
__global__ __launch_bounds__(1024) void tpl_emit_scan_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total, int lvl, int nblocks) {
  const TplJob J = jobs[blockIdx.y];
  int *block_count = tpl_view(J, w, h, px_total).block_count;
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan (integers: order is immaterial)
      const int t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_count[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) J.d_n[lvl] = carry;
}

__global__ __launch_bounds__(kEmitThreads) void tpl_emit_write_kernel(const TplJob *__restrict__ jobs, int w, int h, int px_total, int lvl,
                                                                      int nitems, int wi, int wl, int off, int texel_floats) {
  const TplJob J = jobs[blockIdx.y];
  const TplView V = tpl_view(J, w, h, px_total);
  const float *idl = V.idB + off, *ws = V.wsB + off, *ref = J.ref[lvl];
  const int *block_offset = V.block_count;
  float4 *pts = J.pts[lvl];
  __shared__ int wave_cnt[kEmitItems][kEmitThreads / 64];
  const int base = blockIdx.x * kEmitBlock;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 e[kEmitItems];
  bool v[kEmitItems];
  int before[kEmitItems]; // valid items before this lane inside its wave, for pass j
#pragma unroll
  for (int j = 0; j < kEmitItems; j++) {
    const int item = base + j * kEmitThreads + threadIdx.x;
    v[j] = item < nitems && tpl_emit_item(item, wi, wl, idl, ws, ref, texel_floats, e[j]);
    const unsigned long long m = __ballot(v[j]);
    before[j] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[j][wave] = __builtin_popcountll(m);
  }
  __syncthreads();
  int pos = block_offset[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kEmitItems; j++) {
    int off2 = pos;
    for (int w2 = 0; w2 < wave; w2++) off2 += wave_cnt[j][w2];
    if (v[j]) pts[off2 + before[j]] = e[j];
    pos += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
  }
}

// floats of a job's workspace (everything but its points, which live in the call's point region)
size_t make_coarse_depth_workspace_floats(int w, int h, int nlevels, int npts) {
  size_t px = 0;
  for (int l = 0; l < nlevels; l++) px += (size_t)(w >> l) * (h >> l);
  const size_t blocks = ((size_t)w * h + kEmitBlock - 1) / kEmitBlock + 1;
  return 4 * px + (size_t)w * h + (size_t)npts + blocks + 64;
}

// jobs: device table of njobs entries (ws = make_coarse_depth_workspace_floats() floats each; pt = [pu | pv | pidepth | pweight], npts
// each, on the device; ref[l] = the keyframe's pyramid level l (texel_floats = kTexel: the intensity plane); pts[l] = the float4
// template buffers; d_n = nlevels + 1 ints: n per level, then the out-of-bounds flag).  max_npts = the largest npts of the table.
void launch_make_coarse_depth(hipStream_t s, int w, int h, int nlevels, const TplJob *jobs, int njobs, int max_npts, int texel_floats) {
  if (njobs <= 0) return;
  int px = 0;
  for (int l = 0; l < nlevels; l++) px += (w >> l) * (h >> l);
  const unsigned J = (unsigned)njobs;
  hipLaunchKernelGGL(tpl_init_kernel, dim3((w * h + 255) / 256, J), dim3(256), 0, s, jobs, w, h, nlevels, px);
  if (max_npts > 0) hipLaunchKernelGGL(tpl_splat_link_kernel, dim3((max_npts + 255) / 256, J), dim3(256), 0, s, jobs, w, h, nlevels, px);
  hipLaunchKernelGGL(tpl_splat_sum_kernel, dim3((w * h + 255) / 256, J), dim3(256), 0, s, jobs, w, h, px);
  int off = 0;
  for (int l = 1; l < nlevels; l++) {
    const int prev = off;
    off += (w >> (l - 1)) * (h >> (l - 1));
    const int wl = w >> l, hl = h >> l;
    hipLaunchKernelGGL(tpl_pyr_sum_kernel, dim3((wl * hl + 255) / 256, J), dim3(256), 0, s, jobs, w, h, px, wl, hl, w >> (l - 1), prev, off);
  }
  off = 0;
  for (int l = 0; l < nlevels; l++) {
    const int wl = w >> l, hl = h >> l;
    hipLaunchKernelGGL(tpl_dilate_kernel, dim3((wl * hl + 255) / 256, J), dim3(256), 0, s, jobs, w, h, px, wl, hl, l < 2 ? 1 : 0, off);
    const int wi = wl - 4, hi = hl - 4;
    const int nitems = wi > 0 && hi > 0 ? wi * hi : 0;
    if (nitems > 0) {
      const int nblocks = (nitems + kEmitBlock - 1) / kEmitBlock;
      hipLaunchKernelGGL(tpl_emit_count_kernel, dim3(nblocks, J), dim3(kEmitThreads), 0, s, jobs, w, h, px, l, nitems, wi, wl, off, texel_floats);
      hipLaunchKernelGGL(tpl_emit_scan_kernel, dim3(1, J), dim3(1024), 0, s, jobs, w, h, px, l, nblocks);
      hipLaunchKernelGGL(tpl_emit_write_kernel, dim3(nblocks, J), dim3(kEmitThreads), 0, s, jobs, w, h, px, l, nitems, wi, wl, off, texel_floats);
    }
    off += wl * hl;
  }
}

} // namespace dsm
