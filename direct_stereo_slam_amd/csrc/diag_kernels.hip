// diag_kernels.hip -- measurement aid: read-only streaming bandwidth of this GPU (the practical
// ceiling the eval kernels are compared against in DESIGN.md / bench.py --membw).
#include "dsm_internal.hpp"

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
} // namespace dsm

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
