// diag_kernels.hip -- measurement aid: read-only streaming bandwidth of this GPU (the practical
// ceiling the eval kernels are compared against in DESIGN.md / bench.py --membw).
#include "dsm_internal.hpp"
#include "xwg_sync.hpp"

namespace dsm {
typedef float fvec4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_bw_kernel(const fvec4d *__restrict__ src, size_t n4, float *__restrict__ out) {
  fvec4d acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // four independent 16-byte loads in flight per lane
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const fvec4d a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += (a + b) + (c + d);
  }
  for (; i < n4; i += stride) acc += src[i];
  const float s = (acc.x + acc.y) + (acc.z + acc.w);
  if (s == 123456.789f) out[0] = s; // never true for the test pattern: keeps the loads alive
}
// Same bytes, but every workgroup streams its own contiguous chunk (the access pattern of the eval
// kernels: one chunk of template points + the image rows they land on per workgroup) instead of all
// workgroups advancing through memory side by side.
__global__ __launch_bounds__(256) void read_bw_chunked_kernel(const fvec4d *__restrict__ src, size_t chunk4, float *__restrict__ out) {
  fvec4d acc = {0.f, 0.f, 0.f, 0.f};
  const fvec4d *p = src + (size_t)blockIdx.x * chunk4;
  size_t i = threadIdx.x;
  for (; i + 3 * 256 < chunk4; i += 4 * 256) {
    const fvec4d a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
    acc += (a + b) + (c + d);
  }
  for (; i < chunk4; i += 256) acc += p[i];
  const float s = (acc.x + acc.y) + (acc.z + acc.w);
  if (s == 123456.789f) out[0] = s;
}

// Queue probe (ensure_streams, dsm_capi.hip): a kernel that stays resident for `ticks` of the constant-rate wall clock (bounded: it
// leaves after 2^22 polls whatever the clock says) and an empty one.  A HIP stream is mapped onto one of a few hardware queues by the
// runtime (four by default, shared round robin with every other stream of the process); two streams on ONE queue run their kernels
// one after the other however independent they are -- the probe makes that visible: the empty kernel on stream B finishes while the
// waiting kernel on stream A is still resident only if A and B sit on different queues.
__global__ __launch_bounds__(64) void queue_probe_wait_kernel(long long ticks, int *sink) {
  const long long t0 = wall_clock64();
  int polls = 0;
  while (wall_clock64() - t0 < ticks && polls < (1 << 22)) {
    __builtin_amdgcn_s_sleep(16);
    polls++;
  }
  if (ticks < 0) *sink = polls;
}
__global__ void queue_probe_empty_kernel() {}
void launch_queue_probe_wait(hipStream_t s, long long ticks) { hipLaunchKernelGGL(queue_probe_wait_kernel, dim3(1), dim3(64), 0, s, ticks, nullptr); }
void launch_queue_probe_empty(hipStream_t s) { hipLaunchKernelGGL(queue_probe_empty_kernel, dim3(1), dim3(1), 0, s); }

// Message-passing litmus of the hand-off protocol the eval / LM / queue kernels use between workgroups (xwg_sync.hpp):
// workgroup 2p produces, workgroup 2p + 1 consumes (consecutive workgroups run on different XCDs: b % 8).  Per hand-off k
// the producer stores a 64-float "partial" whose every word is k with device-scope stores, xwg_release(), then adds 1 to
// the pair's ticket; the consumer polls the ticket (device-scope loads), xwg_acquire(), reads the block back with
// load_partial4 -- from an L1 that still holds hand-off k - 1's lines -- and counts every word that is not k; an
// acknowledgement counter lets the producer go on.  A third of the pairs delay the producer by a varying number of cycles,
// another third the consumer: uneven load.  Waits are bounded (error flag instead of a hang).
__global__ __launch_bounds__(64) void xwg_litmus_kernel(float *__restrict__ payload, int *__restrict__ ticket, int *__restrict__ ack, int iters,
                                                        unsigned long long *__restrict__ stale, int *__restrict__ error) {
  const int pair = blockIdx.x >> 1, lane = threadIdx.x;
  const bool producer = (blockIdx.x & 1) == 0;
  float *blk = payload + (size_t)pair * 64;
  int *tk = ticket + 32 * pair, *ak = ack + 32 * pair; // own 128-byte lines
  unsigned long long bad = 0;
  for (int k = 1; k <= iters; k++) {
    if ((pair % 3 == 1 && producer) || (pair % 3 == 2 && !producer))
      for (int d = (k * 7 + pair) & 15; d > 0; d--) __builtin_amdgcn_s_sleep(4); // a varying delay: uneven load
    if (producer) {
      // wait until the consumer has read hand-off k - 1
      for (unsigned spins = 0; __hip_atomic_load(ak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k - 1; spins++)
        if (spins > (1u << 24)) {
          *error = 1;
          return;
        }
      store_partial(blk + lane, (float)k);
      xwg_release();
      if (lane == 0) atomicAdd(tk, 1);
    } else {
      for (unsigned spins = 0; __hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k; spins++)
        if (spins > (1u << 24)) {
          *error = 2;
          return;
        }
      xwg_acquire();
      if (lane < 16) {
        const xwg_fvec4 v = load_partial4(blk + 4 * lane);
        bad += (v.x != (float)k) + (v.y != (float)k) + (v.z != (float)k) + (v.w != (float)k);
      }
      xwg_release(); // (the loads above are complete before the acknowledgement)
      if (lane == 0) __hip_atomic_store(ak, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (bad) atomicAdd(stale, bad);
}
} // namespace dsm

extern "C" int dsm_diag_xwg_litmus(dsm_context *ctx, int pairs, int iters, long long *handoffs_out, long long *stale_words_out) {
  if (!ctx || pairs < 1 || pairs > 512 || iters < 1 || !handoffs_out || !stale_words_out) {
    dsm::set_error("dsm_diag_xwg_litmus: bad argument");
    return DSM_ERR_INVALID;
  }
  DSM_HIP(hipSetDevice(ctx->device));
  float *payload = nullptr;
  int *flags = nullptr; // [ticket: 32 ints per pair][ack: 32 ints per pair][error][pad]
  unsigned long long *stale = nullptr;
  const size_t nflags = 64 * (size_t)pairs + 32;
  DSM_HIP(hipMalloc(&payload, sizeof(float) * 64 * pairs));
  DSM_HIP(hipMalloc(&flags, sizeof(int) * nflags));
  DSM_HIP(hipMalloc(&stale, 64));
  DSM_HIP(hipMemsetAsync(payload, 0, sizeof(float) * 64 * pairs, ctx->stream));
  DSM_HIP(hipMemsetAsync(flags, 0, sizeof(int) * nflags, ctx->stream));
  DSM_HIP(hipMemsetAsync(stale, 0, 64, ctx->stream));
  // 2 * pairs workgroups of one wave each: all co-resident (<= 1024 waves on 256 CUs), which the spinning needs
  hipLaunchKernelGGL(dsm::xwg_litmus_kernel, dim3(2 * pairs), dim3(64), 0, ctx->stream, payload, flags, flags + 32 * pairs, iters, stale,
                     flags + 64 * pairs);
  unsigned long long h_stale = 0;
  int h_err = 0;
  DSM_HIP(hipMemcpyAsync(&h_stale, stale, sizeof h_stale, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipMemcpyAsync(&h_err, flags + 64 * pairs, sizeof h_err, hipMemcpyDeviceToHost, ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  DSM_HIP(hipFree(payload));
  DSM_HIP(hipFree(flags));
  DSM_HIP(hipFree(stale));
  if (h_err) {
    dsm::set_error("dsm_diag_xwg_litmus: a bounded wait expired (workgroups not co-resident?)");
    return DSM_ERR_STATE;
  }
  *handoffs_out = (long long)pairs * iters;
  *stale_words_out = (long long)h_stale;
  return DSM_OK;
}

extern "C" int dsm_diag_read_bandwidth_chunked(dsm_context *ctx, size_t bytes, size_t chunk_bytes, int iters, double *gbps_out) {
  if (!ctx || !gbps_out || bytes < (1u << 20) || iters < 1 || chunk_bytes < 4096 || chunk_bytes % 16) {
    dsm::set_error("dsm_diag_read_bandwidth_chunked: bad argument");
    return DSM_ERR_INVALID;
  }
  DSM_HIP(hipSetDevice(ctx->device));
  float *buf = nullptr, *out = nullptr;
  const size_t nchunks = bytes / chunk_bytes;
  bytes = nchunks * chunk_bytes;
  DSM_HIP(hipMalloc(&buf, bytes));
  DSM_HIP(hipMalloc(&out, 64));
  DSM_HIP(hipMemsetAsync(buf, 0x3c, bytes, ctx->stream));
  for (int i = -1; i < iters; i++) {
    if (i == 0) DSM_HIP(hipEventRecord(ctx->ev_total[0], ctx->stream));
    hipLaunchKernelGGL(dsm::read_bw_chunked_kernel, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, (const dsm::fvec4d *)buf,
                       chunk_bytes / 16, out);
  }
  DSM_HIP(hipEventRecord(ctx->ev_total[1], ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  DSM_HIP(hipEventElapsedTime(&ms, ctx->ev_total[0], ctx->ev_total[1]));
  *gbps_out = (double)bytes * iters / (ms * 1e-3) / 1e9;
  DSM_HIP(hipFree(buf));
  DSM_HIP(hipFree(out));
  return DSM_OK;
}

extern "C" int dsm_diag_read_bandwidth(dsm_context *ctx, size_t bytes, int iters, double *gbps_out) {
  if (!ctx || !gbps_out || bytes < (1u << 20) || iters < 1) {
    dsm::set_error("dsm_diag_read_bandwidth: bad argument");
    return DSM_ERR_INVALID;
  }
  DSM_HIP(hipSetDevice(ctx->device));
  float *buf = nullptr, *out = nullptr;
  DSM_HIP(hipMalloc(&buf, bytes));
  DSM_HIP(hipMalloc(&out, 64));
  DSM_HIP(hipMemsetAsync(buf, 0x3c, bytes, ctx->stream));
  const size_t n4 = bytes / 16;
  const int grid = 256 * 8;
  hipLaunchKernelGGL(dsm::read_bw_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const dsm::fvec4d *)buf, n4, out);
  DSM_HIP(hipEventRecord(ctx->ev_total[0], ctx->stream));
  for (int i = 0; i < iters; i++)
    hipLaunchKernelGGL(dsm::read_bw_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const dsm::fvec4d *)buf, n4, out);
  DSM_HIP(hipEventRecord(ctx->ev_total[1], ctx->stream));
  DSM_HIP(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  DSM_HIP(hipEventElapsedTime(&ms, ctx->ev_total[0], ctx->ev_total[1]));
  *gbps_out = (double)bytes * iters / (ms * 1e-3) / 1e9;
  DSM_HIP(hipFree(buf));
  DSM_HIP(hipFree(out));
  return DSM_OK;
}
