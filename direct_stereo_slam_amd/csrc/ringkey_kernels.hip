// ringkey_kernels.hip -- exact brute-force k-NN over ScanContext ring keys (gfx950).
//
// Replaces the flann::Index<flann::L2<float>> KD-forest queried by search_ringkey
// (src/loop_closure/loop_detection/search_place.h:25-39, built at LoopHandler.cpp:35-39) with an
// exact scan.  Distances follow flann::L2<float>::operator() (squared L2, accumulated in groups
// of four, UPSTREAM FLANN dist.h) with -ffp-contract=off, so they are bit-identical to the CPU
// oracle; candidates are packed as  (float_bits(dist2) << 32) | global_index  so that an unsigned
// 64-bit min is "smaller distance first, smaller index on ties" -- the merge order used inside
// the kernel, across slices, and across GPUs (RCCL all-reduce(min)).
//
// HBM layout: dimension-major planes keysT[j * capacity + i] (i = local slot of this shard), so the
// tile loads are fully coalesced; global index = i * shard_count + shard_rank.
#include "dsm_kernels.hpp"

namespace dsm {

constexpr int kRkThreads = 256;
constexpr int kRkTile = 128;
// "no candidate": larger than every packed candidate both as uint64 and as int64 (dist2 >= 0 keeps bit 63 clear)
constexpr unsigned long long kNoCand = 0x7FFFFFFFFFFFFFFFull;

template <int K>
__device__ __forceinline__ void topk_insert(unsigned long long (&t)[K], unsigned long long c) {
#pragma unroll
  for (int j = 0; j < K; j++) {
    const bool lt = c < t[j];
    const unsigned long long lo = lt ? c : t[j];
    c = lt ? t[j] : c;
    t[j] = lo;
  }
}

// A thread visits its keys in ascending index order, so a candidate at the distance of the list's K-th entry can never
// displace it (same distance, larger index): once the list is full only a STRICTLY smaller distance gets in.  `cut` is
// that bound -- thres while the list has room, the K-th distance after -- and the per-pair test `result < cut` replaces
// `result < thres`: the same lists, but the 64-bit compare-and-swap chain (20 vector instructions) runs for the handful of
// candidates that change a list instead of for every pair under the threshold (a third of all pairs on keys of similar
// scenes, where it was a quarter of the many-query kernel's instructions).
template <int K>
__device__ __forceinline__ void topk_insert_cut(unsigned long long (&t)[K], float &cut, float result, unsigned long long g) {
  topk_insert<K>(t, ((unsigned long long)__float_as_uint(result) << 32) | g);
  if (t[K - 1] != kNoCand) cut = __uint_as_float((unsigned)(t[K - 1] >> 32));
}

// one thread = QPT queries; key tiles are staged through LDS and broadcast to all lanes.  QPT = 2 (large query counts, dim
// 20): every broadcast read of a key element feeds two subtractions -- a ds_read_b128 costs the CU's one LDS four cycles
// per wave whatever the lanes read, five of them per key and wave against 118 vector cycles per SIMD for the 59 operations of
// a pair: with one query per thread the LDS is two thirds busy and the vector pipes wait on it.
template <int DIM, int K, int QPT = 1>
__global__ __launch_bounds__(kRkThreads) void ringkey_knn_kernel(const float *__restrict__ keysT, long long cap,
                                                                 long long n_local, int dim_rt, float thres,
                                                                 int shard_rank, int shard_count,
                                                                 const float *__restrict__ queries, int nq,
                                                                 int n_slices,
                                                                 unsigned long long *__restrict__ scratch) {
  constexpr int DMAX = DIM > 0 ? DIM : 32;
  static_assert(QPT == 1 || DIM > 0, "several queries per thread: fixed dimension only");
  const int dim = DIM > 0 ? DIM : dim_rt;
  __shared__ __attribute__((aligned(16))) float tile[kRkTile * DMAX];
  int q[QPT];
#pragma unroll
  for (int i = 0; i < QPT; i++) q[i] = (blockIdx.x * QPT + i) * kRkThreads + threadIdx.x;
  const int slice = blockIdx.y;
  const long long per = (n_local + n_slices - 1) / n_slices;
  const long long k0 = (long long)slice * per;
  const long long k1 = k0 + per < n_local ? k0 + per : n_local;

  float qv[QPT][DMAX];
  if (DIM > 0) {
#pragma unroll
    for (int i = 0; i < QPT; i++)
#pragma unroll
      for (int j = 0; j < DMAX; j++) qv[i][j] = q[i] < nq ? queries[(size_t)q[i] * DIM + j] : 0.f;
  }
  unsigned long long best[QPT][K];
  float cut[QPT];
#pragma unroll
  for (int i = 0; i < QPT; i++) {
    cut[i] = q[i] < nq ? thres : -1.0f; // (a thread without a query never passes the test)
#pragma unroll
    for (int j = 0; j < K; j++) best[i][j] = kNoCand;
  }

  for (long long base = k0; base < k1; base += kRkTile) {
    const int tn = (int)(k1 - base < kRkTile ? k1 - base : kRkTile);
    __syncthreads();
    for (int e = threadIdx.x; e < dim * kRkTile; e += kRkThreads) {
      const int j = e / kRkTile, t = e % kRkTile;
      if (t < tn) tile[t * dim + j] = keysT[(size_t)j * cap + base + t];
    }
    __syncthreads();
    if (q[0] < nq) {
#pragma unroll 4 // (four keys' LDS reads in flight per wait)
      for (int t = 0; t < tn; t++) {
        const float *kp = tile + t * dim;
        float result[QPT];
#pragma unroll
        for (int i = 0; i < QPT; i++) result[i] = 0.f;
        if (DIM > 0) {
#pragma unroll
          for (int j = 0; j < DMAX; j += 4) { // flann::L2 main loop
            const float k0v = kp[j], k1v = kp[j + 1], k2v = kp[j + 2], k3v = kp[j + 3];
#pragma unroll
            for (int i = 0; i < QPT; i++) {
              const float d0 = qv[i][j] - k0v, d1 = qv[i][j + 1] - k1v, d2 = qv[i][j + 2] - k2v, d3 = qv[i][j + 3] - k3v;
              result[i] += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
          }
        } else {
          const float *qp = queries + (size_t)q[0] * dim;
          int j = 0;
          for (; j + 3 < dim; j += 4) {
            const float d0 = qp[j] - kp[j], d1 = qp[j + 1] - kp[j + 1], d2 = qp[j + 2] - kp[j + 2],
                        d3 = qp[j + 3] - kp[j + 3];
            result[0] += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          }
          for (; j < dim; j++) { // flann::L2 tail loop
            const float d0 = qp[j] - kp[j];
            result[0] += d0 * d0;
          }
        }
#pragma unroll
        for (int i = 0; i < QPT; i++)
          if (result[i] < cut[i]) topk_insert_cut<K>(best[i], cut[i], result[i], (unsigned long long)(base + t) * shard_count + shard_rank);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < QPT; i++)
    if (q[i] < nq) {
#pragma unroll
      for (int j = 0; j < K; j++) scratch[((size_t)slice * nq + q[i]) * K + j] = best[i][j];
    }
}

// Few queries (the SLAM case: one ring key per keyframe, a handful with several concurrent sequences): one thread =
// one key, the queries of a group of kRkQG sit in LDS and are broadcast, every thread keeps a top-K per query over the
// keys it visits, the workgroup merges them at the end.  One sweep of the key planes per query group, fully coalesced:
// HBM-bound (80 bytes per key) instead of one busy lane per query.  Same per-pair distance expression, same packed
// candidates, same scratch layout as ringkey_knn_kernel -- bit-identical results.
constexpr int kRkQG = 8; // largest query group

template <int DIM, int K, int QG>
__global__ __launch_bounds__(kRkThreads) void ringkey_knn_fewq_kernel(const float *__restrict__ keysT, long long cap, long long n_local,
                                                                      float thres, int shard_rank, int shard_count,
                                                                      const float *__restrict__ queries, int nq, int n_slices,
                                                                      unsigned long long *__restrict__ scratch) {
  static_assert(DIM % 4 == 0, "flann::L2 main loop only");
  __shared__ __attribute__((aligned(16))) float qs[QG][DIM];
  __shared__ unsigned long long wtop[kRkThreads / 64][QG][K];
  const int slice = blockIdx.x, q0 = blockIdx.y * QG;
  const int nqg = nq - q0 < QG ? nq - q0 : QG;
  const long long per = (n_local + n_slices - 1) / n_slices;
  const long long k0 = (long long)slice * per;
  const long long k1 = k0 + per < n_local ? k0 + per : n_local;
  for (int e = threadIdx.x; e < QG * DIM; e += kRkThreads) {
    const int qq = e / DIM, j = e % DIM;
    qs[qq][j] = qq < nqg ? queries[(size_t)(q0 + qq) * DIM + j] : 0.f;
  }
  __syncthreads();
  unsigned long long best[QG][K];
  float cut[QG];
#pragma unroll
  for (int qq = 0; qq < QG; qq++) {
    cut[qq] = qq < nqg ? thres : -1.0f;
#pragma unroll
    for (int j = 0; j < K; j++) best[qq][j] = kNoCand;
  }
  for (long long i = k0 + threadIdx.x; i < k1; i += kRkThreads) {
    float kv[DIM];
#pragma unroll
    for (int j = 0; j < DIM; j++) kv[j] = keysT[(size_t)j * cap + i];
    const unsigned long long g = (unsigned long long)i * shard_count + shard_rank;
#pragma unroll
    for (int qq = 0; qq < QG; qq++) {
      float result = 0.f;
#pragma unroll
      for (int j = 0; j < DIM; j += 4) { // flann::L2 main loop
        const float d0 = qs[qq][j] - kv[j], d1 = qs[qq][j + 1] - kv[j + 1], d2 = qs[qq][j + 2] - kv[j + 2],
                    d3 = qs[qq][j + 3] - kv[j + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
      if (result < cut[qq]) topk_insert_cut<K>(best[qq], cut[qq], result, g);
    }
  }
  // per wave and query: K rounds of wave-min extraction (the winner lane pops its head) -> LDS; then one thread per
  // query merges the waves' lists
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int qq = 0; qq < QG; qq++) {
#pragma unroll
    for (int r = 0; r < K; r++) {
      unsigned long long m = best[qq][0];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 64);
        m = o < m ? o : m;
      }
      if (lane == 0) wtop[wave][qq][r] = m;
      if (m != kNoCand && best[qq][0] == m) { // packed candidates are unique (global index in the low bits)
#pragma unroll
        for (int j = 0; j + 1 < K; j++) best[qq][j] = best[qq][j + 1];
        best[qq][K - 1] = kNoCand;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nqg) {
    const int qq = threadIdx.x;
    unsigned long long t[K];
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = kNoCand;
    for (int w = 0; w < kRkThreads / 64; w++)
#pragma unroll
      for (int j = 0; j < K; j++)
        if (wtop[w][qq][j] != kNoCand) topk_insert<K>(t, wtop[w][qq][j]);
#pragma unroll
    for (int j = 0; j < K; j++) scratch[((size_t)slice * nq + q0 + qq) * K + j] = t[j];
  }
}

// The same scan with FOUR consecutive keys per thread and iteration: one 16-byte load per plane and thread, i.e. 1 KiB
// contiguous per plane and wave instead of 256 bytes -- twenty interleaved streams of 256-byte pieces reach 4.5 TB/s on
// this memory system, 1 KiB pieces more (round 3: DESIGN.md section 4.5).  For groups of one or two queries (the registers
// hold 80 key values); same distances, same packed candidates: bit-identical.
typedef float rk_fvec4 __attribute__((ext_vector_type(4)));
template <int DIM, int K, int QG>
__global__ __launch_bounds__(kRkThreads) void ringkey_knn_fewq4_kernel(const float *__restrict__ keysT, long long cap, long long n_local,
                                                                       float thres, int shard_rank, int shard_count,
                                                                       const float *__restrict__ queries, int nq, int n_slices,
                                                                       unsigned long long *__restrict__ scratch) {
  static_assert(DIM % 4 == 0, "flann::L2 main loop only");
  __shared__ __attribute__((aligned(16))) float qs[QG][DIM];
  __shared__ unsigned long long wtop[kRkThreads / 64][QG][K];
  const int slice = blockIdx.x, q0 = blockIdx.y * QG;
  const int nqg = nq - q0 < QG ? nq - q0 : QG;
  const long long per = (((n_local + n_slices - 1) / n_slices) + 3) & ~3ll; // slices start on 16-byte boundaries of the planes
  const long long k0 = (long long)slice * per;
  const long long k1 = k0 + per < n_local ? k0 + per : n_local;
  for (int e = threadIdx.x; e < QG * DIM; e += kRkThreads) {
    const int qq = e / DIM, j = e % DIM;
    qs[qq][j] = qq < nqg ? queries[(size_t)(q0 + qq) * DIM + j] : 0.f;
  }
  __syncthreads();
  unsigned long long best[QG][K];
  float cut[QG];
#pragma unroll
  for (int qq = 0; qq < QG; qq++) {
    cut[qq] = qq < nqg ? thres : -1.0f;
#pragma unroll
    for (int j = 0; j < K; j++) best[qq][j] = kNoCand;
  }
  for (long long i = k0 + 4 * threadIdx.x; i < k1; i += 4 * kRkThreads) {
    rk_fvec4 kv[DIM];
#pragma unroll
    for (int j = 0; j < DIM; j++) kv[j] = __builtin_nontemporal_load((const rk_fvec4 *)(keysT + (size_t)j * cap + i));
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const unsigned long long g = (unsigned long long)(i + e) * shard_count + shard_rank;
#pragma unroll
      for (int qq = 0; qq < QG; qq++) {
        float result = 0.f;
#pragma unroll
        for (int j = 0; j < DIM; j += 4) { // flann::L2 main loop
          const float d0 = qs[qq][j] - kv[j][e], d1 = qs[qq][j + 1] - kv[j + 1][e], d2 = qs[qq][j + 2] - kv[j + 2][e],
                      d3 = qs[qq][j + 3] - kv[j + 3][e];
          result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        if (i + e < k1 && result < cut[qq]) topk_insert_cut<K>(best[qq], cut[qq], result, g);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int qq = 0; qq < QG; qq++) {
#pragma unroll
    for (int r = 0; r < K; r++) {
      unsigned long long m = best[qq][0];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 64);
        m = o < m ? o : m;
      }
      if (lane == 0) wtop[wave][qq][r] = m;
      if (m != kNoCand && best[qq][0] == m) {
#pragma unroll
        for (int j = 0; j + 1 < K; j++) best[qq][j] = best[qq][j + 1];
        best[qq][K - 1] = kNoCand;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nqg) {
    const int qq = threadIdx.x;
    unsigned long long t[K];
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = kNoCand;
    for (int w = 0; w < kRkThreads / 64; w++)
#pragma unroll
      for (int j = 0; j < K; j++)
        if (wtop[w][qq][j] != kNoCand) topk_insert<K>(t, wtop[w][qq][j]);
#pragma unroll
    for (int j = 0; j < K; j++) scratch[((size_t)slice * nq + q0 + qq) * K + j] = t[j];
  }
}

// one wave per query merges the per-slice candidates
template <int K>
__global__ __launch_bounds__(64) void ringkey_merge_kernel(const unsigned long long *__restrict__ scratch, int nq,
                                                           int n_slices, unsigned long long *__restrict__ out) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  unsigned long long best[K];
#pragma unroll
  for (int j = 0; j < K; j++) best[j] = kNoCand;
  for (int s = lane; s < n_slices; s += 64) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      const unsigned long long c = scratch[((size_t)s * nq + q) * K + j];
      if (c != kNoCand) topk_insert<K>(best, c);
    }
  }
  // K rounds of wave-min extraction; the winner lane pops its head
#pragma unroll
  for (int r = 0; r < K; r++) {
    unsigned long long m = best[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off, 64);
      m = o < m ? o : m;
    }
    if (lane == 0) out[(size_t)q * K + r] = m;
    if (m != kNoCand && best[0] == m) {
#pragma unroll
      for (int j = 0; j + 1 < K; j++) best[j] = best[j + 1];
      best[K - 1] = kNoCand;
    }
  }
}

__global__ void ringkey_insert_kernel(float *keysT, long long cap, long long pos, int dim, const float *key,
                                      int nkeys) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < dim * nkeys) {
    const int kk = e / dim, j = e % dim;
    keysT[(size_t)j * cap + pos + kk] = key[(size_t)kk * dim + j];
  }
}

constexpr int kRkTwoPerThread = 512; // from here on a thread of ringkey_knn_kernel carries two queries
constexpr int kRkFewQueries = 32; // up to here the thread-per-key kernel is used (dim 20 only)
static bool ringkey_use_fewq(int dim, int nq) { return dim == 20 && nq <= kRkFewQueries; }

int ringkey_num_slices(int64_t n_local, int nq, int dim) {
  if (ringkey_use_fewq(dim, nq)) {
    // about eight workgroups per CU for one query group, at least 1024 keys (four per thread) per workgroup
    int64_t s = (n_local + 1023) / 1024;
    if (s > 2048) s = 2048;
    return (int)(s < 1 ? 1 : s);
  }
  // enough workgroups to fill 256 CUs, at least one key tile per slice
  const int per_block = dim == 20 && nq >= kRkTwoPerThread ? 2 * kRkThreads : kRkThreads;
  const int qblocks = (nq + per_block - 1) / per_block;
  // one full round of resident workgroups: 8 per CU for the one-query-per-thread kernel (54 VGPRs), 5 for two per thread (88):
  // 1.6 rounds cost two
  const int resident = per_block == kRkThreads ? 2048 : 1280;
  int64_t want = resident / (qblocks > 0 ? qblocks : 1);
  if (want < 1) want = 1;
  const int64_t max_by_keys = (n_local + kRkTile - 1) / kRkTile;
  int64_t s = want < max_by_keys ? want : max_by_keys;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  return (int)s;
}

template <int K>
static void launch_knn_k(hipStream_t s, const float *keysT, int64_t cap, int64_t n_local, int dim, float thres,
                         int shard_rank, int shard_count, const float *d_queries, int nq,
                         unsigned long long *d_scratch, int n_slices, unsigned long long *d_packed_out) {
  dim3 grid((nq + kRkThreads - 1) / kRkThreads, n_slices), block(kRkThreads);
  if (ringkey_use_fewq(dim, nq)) {
#define DSM_FEWQ(QG)                                                                                                   \
  hipLaunchKernelGGL((ringkey_knn_fewq_kernel<20, K, QG>), dim3(n_slices, (nq + QG - 1) / QG), block, 0, s, keysT, (long long)cap,       \
                     (long long)n_local, thres, shard_rank, shard_count, d_queries, nq, n_slices, d_scratch)
#define DSM_FEWQ4(QG)                                                                                                  \
  hipLaunchKernelGGL((ringkey_knn_fewq4_kernel<20, K, QG>), dim3(n_slices, (nq + QG - 1) / QG), block, 0, s, keysT, (long long)cap,      \
                     (long long)n_local, thres, shard_rank, shard_count, d_queries, nq, n_slices, d_scratch)
    if (nq == 1 && (cap & 3) == 0)
      DSM_FEWQ4(1);
    else if (nq == 2 && (cap & 3) == 0)
      DSM_FEWQ4(2);
    else if (nq == 1) // (groups of four / eight queries with four keys per thread: 408 us against 274 for eight queries over 10^7 keys -- registers)
      DSM_FEWQ(1);
    else if (nq == 2)
      DSM_FEWQ(2);
    else if (nq <= 4)
      DSM_FEWQ(4);
    else
      DSM_FEWQ(kRkQG);
#undef DSM_FEWQ
#undef DSM_FEWQ4
  } else if (dim == 20 && nq >= kRkTwoPerThread)
    hipLaunchKernelGGL((ringkey_knn_kernel<20, K, 2>), dim3((nq + 2 * kRkThreads - 1) / (2 * kRkThreads), n_slices), block, 0, s, keysT,
                       (long long)cap, (long long)n_local, dim, thres, shard_rank, shard_count, d_queries, nq, n_slices, d_scratch);
  else if (dim == 20)
    hipLaunchKernelGGL((ringkey_knn_kernel<20, K>), grid, block, 0, s, keysT, (long long)cap, (long long)n_local, dim,
                       thres, shard_rank, shard_count, d_queries, nq, n_slices, d_scratch);
  else
    hipLaunchKernelGGL((ringkey_knn_kernel<0, K>), grid, block, 0, s, keysT, (long long)cap, (long long)n_local, dim,
                       thres, shard_rank, shard_count, d_queries, nq, n_slices, d_scratch);
  hipLaunchKernelGGL((ringkey_merge_kernel<K>), dim3(nq), dim3(64), 0, s, d_scratch, nq, n_slices, d_packed_out);
}

void launch_ringkey_knn(hipStream_t s, const float *keysT, int64_t cap, int64_t n_local, int dim, int k, float thres,
                        int shard_rank, int shard_count, const float *d_queries, int nq,
                        unsigned long long *d_scratch, int n_slices, unsigned long long *d_packed_out) {
  switch (k) {
  case 1: launch_knn_k<1>(s, keysT, cap, n_local, dim, thres, shard_rank, shard_count, d_queries, nq, d_scratch, n_slices, d_packed_out); break;
  case 2: launch_knn_k<2>(s, keysT, cap, n_local, dim, thres, shard_rank, shard_count, d_queries, nq, d_scratch, n_slices, d_packed_out); break;
  case 3: launch_knn_k<3>(s, keysT, cap, n_local, dim, thres, shard_rank, shard_count, d_queries, nq, d_scratch, n_slices, d_packed_out); break;
  default: launch_knn_k<4>(s, keysT, cap, n_local, dim, thres, shard_rank, shard_count, d_queries, nq, d_scratch, n_slices, d_packed_out); break;
  }
}

void launch_ringkey_insert(hipStream_t s, float *keysT, int64_t cap, int64_t pos, int dim, const float *d_key,
                           int nkeys) {
  const int n = dim * nkeys;
  hipLaunchKernelGGL(ringkey_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keysT, (long long)cap,
                     (long long)pos, dim, d_key, nkeys);
}

} // namespace dsm
