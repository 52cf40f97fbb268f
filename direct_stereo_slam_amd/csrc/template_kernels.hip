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

__global__ void tpl_splat_link_kernel(int npts, const float *__restrict__ pu, const float *__restrict__ pv, int w, int h,
                                      int *__restrict__ head, int *__restrict__ next, int *__restrict__ err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npts) return;
  const int u = (int)(pu[k] + 0.5f); // :151-152
  const int v = (int)(pv[k] + 0.5f);
  if (u < 0 || v < 0 || u >= w || v >= h) { // the reference would write out of bounds
    atomicOr(err, 1);
    next[k] = -2;
    return;
  }
  next[k] = atomicExch(&head[u + w * v], k);
}

// one thread per level-0 pixel: sum this pixel's points in ascending point index (:160-161)
__global__ void tpl_splat_sum_kernel(int npix, const int *__restrict__ head, const int *__restrict__ next,
                                     const float *__restrict__ pidepth, const float *__restrict__ pweight,
                                     float *__restrict__ idepth0, float *__restrict__ wsum0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  float sid = 0.f, sw = 0.f;
  const int hd = head[p];
  int last = -1;
  while (hd >= 0) {
    int best = 0x7FFFFFFF;
    for (int j = hd; j >= 0; j = next[j])
      if (j > last && j < best) best = j;
    if (best == 0x7FFFFFFF) break;
    sid += pidepth[best] * pweight[best];
    sw += pweight[best];
    last = best;
  }
  idepth0[p] = sid;
  wsum0[p] = sw;
}

__global__ void tpl_pyr_sum_kernel(int wl, int hl, int wlm1, const float *__restrict__ idm, const float *__restrict__ wsm,
                                   float *__restrict__ idl, float *__restrict__ wsl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wl * hl) return;
  const int x = i % wl, y = i / wl;
  const int b = 2 * x + 2 * y * wlm1;
  idl[i] = idm[b] + idm[b + 1] + idm[b + wlm1] + idm[b + wlm1 + 1];
  wsl[i] = wsm[b] + wsm[b + 1] + wsm[b + wlm1] + wsm[b + wlm1 + 1];
}

// :190-233 (levels 0-1, diagonal neighbours) / :236-275 (levels >= 2, axis neighbours)
__global__ void tpl_dilate_kernel(int wl, int hl, int diagonal, const float *__restrict__ idl_in, const float *__restrict__ bak,
                                  float *__restrict__ idl_out, float *__restrict__ ws_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int npix = wl * hl;
  if (i >= npix) return;
  float id = idl_in[i], ws = bak[i];
  if (i >= wl && i < npix - wl && ws <= 0) {
    const int o0 = diagonal ? 1 + wl : 1, o1 = diagonal ? -1 - wl : -1, o2 = diagonal ? wl - 1 : wl, o3 = diagonal ? -wl + 1 : -wl;
    const int off[4] = {o0, o1, o2, o3};
    float sum = 0, num = 0, numn = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (bak[i + off[k]] > 0) {
        sum += idl_in[i + off[k]];
        num += bak[i + off[k]];
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

__global__ __launch_bounds__(kEmitThreads) void tpl_emit_count_kernel(int nitems, int wi, int wl, const float *__restrict__ idl,
                                                                      const float *__restrict__ ws, const float *__restrict__ ref,
                                                                      int texel_floats, int *__restrict__ block_count) {
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
  if (threadIdx.x == 0) block_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// exclusive scan of the block counts by one workgroup (<= 1024 counts per pass, carried), total -> *n_out
__global__ __launch_bounds__(1024) void tpl_emit_scan_kernel(int nblocks, int *__restrict__ block_count, int *__restrict__ n_out) {
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
  if (threadIdx.x == 0) *n_out = carry;
}

__global__ __launch_bounds__(kEmitThreads) void tpl_emit_write_kernel(int nitems, int wi, int wl, const float *__restrict__ idl,
                                                                      const float *__restrict__ ws, const float *__restrict__ ref,
                                                                      int texel_floats, const int *__restrict__ block_offset,
                                                                      float4 *__restrict__ pts) {
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
    int off = pos;
    for (int w2 = 0; w2 < wave; w2++) off += wave_cnt[j][w2];
    if (v[j]) pts[off + before[j]] = e[j];
    pos += wave_cnt[j][0] + wave_cnt[j][1] + wave_cnt[j][2] + wave_cnt[j][3];
  }
}

size_t make_coarse_depth_workspace_floats(int w, int h, int nlevels, int npts) {
  size_t px = 0;
  for (int l = 0; l < nlevels; l++) px += (size_t)(w >> l) * (h >> l);
  const size_t blocks = ((size_t)w * h + kEmitBlock - 1) / kEmitBlock + 1;
  return 4 * px + (size_t)w * h + 5 * (size_t)npts + blocks + 64;
}

// ws: workspace of make_coarse_depth_workspace_floats() floats.  d_pt: [pu | pv | pidepth | pweight] (npts each), already
// on the device at the start of ws.  ref[l]: the keyframe's pyramid level l (texel_floats = kTexel: the intensity plane).  pts[l]: float4
// template buffers.  d_n: nlevels + 1 ints on the device: n per level, then the out-of-bounds flag.
void launch_make_coarse_depth(hipStream_t s, int w, int h, int nlevels, int npts, float *ws, const float *const *ref,
                              int texel_floats, float4 *const *pts, int *d_n) {
  float *pu = ws, *pv = pu + npts, *pid = pv + npts, *pw = pid + npts;
  int *next = (int *)(pw + npts);
  int *head = next + npts;
  float *lvl_base = (float *)(head + (size_t)w * h);
  size_t px = 0;
  for (int l = 0; l < nlevels; l++) px += (size_t)(w >> l) * (h >> l);
  float *idA = lvl_base, *wsA = idA + px, *idB = wsA + px, *wsB = idB + px;
  int *block_count = (int *)(wsB + px);
  int *err = d_n + nlevels;
  hipMemsetAsync(head, 0xFF, sizeof(int) * (size_t)w * h, s);
  hipMemsetAsync(d_n, 0, sizeof(int) * (nlevels + 1), s);
  if (npts > 0) hipLaunchKernelGGL(tpl_splat_link_kernel, dim3((npts + 255) / 256), dim3(256), 0, s, npts, pu, pv, w, h, head, next, err);
  hipLaunchKernelGGL(tpl_splat_sum_kernel, dim3((w * h + 255) / 256), dim3(256), 0, s, w * h, head, next, pid, pw, idA, wsA);
  size_t off = 0;
  for (int l = 1; l < nlevels; l++) {
    const size_t prev = off;
    off += (size_t)(w >> (l - 1)) * (h >> (l - 1));
    const int wl = w >> l, hl = h >> l;
    hipLaunchKernelGGL(tpl_pyr_sum_kernel, dim3((wl * hl + 255) / 256), dim3(256), 0, s, wl, hl, w >> (l - 1), idA + prev, wsA + prev,
                       idA + off, wsA + off);
  }
  off = 0;
  for (int l = 0; l < nlevels; l++) {
    const int wl = w >> l, hl = h >> l;
    hipLaunchKernelGGL(tpl_dilate_kernel, dim3((wl * hl + 255) / 256), dim3(256), 0, s, wl, hl, l < 2 ? 1 : 0, idA + off, wsA + off,
                       idB + off, wsB + off);
    const int wi = wl - 4, hi = hl - 4;
    const int nitems = wi > 0 && hi > 0 ? wi * hi : 0;
    if (nitems > 0) {
      const int nblocks = (nitems + kEmitBlock - 1) / kEmitBlock;
      hipLaunchKernelGGL(tpl_emit_count_kernel, dim3(nblocks), dim3(kEmitThreads), 0, s, nitems, wi, wl, idB + off, wsB + off, ref[l],
                         texel_floats, block_count);
      hipLaunchKernelGGL(tpl_emit_scan_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_count, d_n + l);
      hipLaunchKernelGGL(tpl_emit_write_kernel, dim3(nblocks), dim3(kEmitThreads), 0, s, nitems, wi, wl, idB + off, wsB + off, ref[l],
                         texel_floats, block_count, pts[l]);
    }
    off += (size_t)wl * hl;
  }
}

} // namespace dsm
