// ringdb_capi.hip -- C ABI of the ring-key database: replaces the flann::Index created at
// LoopHandler.cpp:35-39 and the function-static delay queue of search_ringkey
// (search_place.h:41-56).  Host code keeps the queue and the ordinal bookkeeping; the scan runs
// in ringkey_kernels.hip.
#include <cstdint>
#include <cstring>
#include <vector>

#include "dsm_internal.hpp"
#include <string>

#include "ringdb_internal.hpp"

using namespace dsm;

static int invalid(const char *m) {
  set_error(m);
  return DSM_ERR_INVALID;
}

static int rdb_reserve(dsm_ringdb *db, int64_t need_local) {
  if (need_local <= db->cap) return DSM_OK;
  int64_t ncap = db->cap > 0 ? db->cap : 1024;
  while (ncap < need_local) ncap *= 2;
  float *nk = nullptr;
  DSM_HIP(hipMalloc(&nk, sizeof(float) * (size_t)ncap * db->dim));
  hipError_t e = hipSuccess;
  if (db->d_keysT && db->n_local > 0)
    for (int j = 0; j < db->dim && e == hipSuccess; j++)
      e = hipMemcpyAsync(nk + (size_t)j * ncap, db->d_keysT + (size_t)j * db->cap, sizeof(float) * db->n_local,
                         hipMemcpyDeviceToDevice, db->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(db->ctx->stream);
  if (e != hipSuccess) { // the old planes stay valid; give the new allocation back
    hipFree(nk);
    DSM_HIP(e);
  }
  if (db->d_keysT) DSM_HIP(hipFree(db->d_keysT));
  db->d_keysT = nk;
  db->cap = ncap;
  return DSM_OK;
}

static int rdb_stage(dsm_ringdb *db, size_t floats) {
  if (floats <= db->q_floats) return DSM_OK;
  if (db->d_q) DSM_HIP(hipFree(db->d_q));
  db->d_q = nullptr;
  DSM_HIP(hipMalloc(&db->d_q, floats * sizeof(float)));
  db->q_floats = floats;
  return DSM_OK;
}

// append n keys with global ordinals size_global .. size_global+n-1; keep those of this shard
static int rdb_append(dsm_ringdb *db, const float *keys, int64_t n) {
  // candidates are packed (float_bits(d^2) << 32) | global index, and dsm_ringdb_query_then_enqueue hands indices out
  // as int (search_place.h:36): the global index must stay below 2^31
  if (db->size_global + n > (int64_t)INT32_MAX) return invalid("ring-key index full: global ordinals are limited to 2^31 - 1");
  std::vector<float> mine;
  mine.reserve((size_t)(n / db->shard_count + 1) * db->dim);
  for (int64_t i = 0; i < n; i++) {
    const int64_t g = db->size_global + i;
    if (g % db->shard_count == db->shard_rank) mine.insert(mine.end(), keys + i * db->dim, keys + (i + 1) * db->dim);
  }
  const int64_t m = (int64_t)mine.size() / db->dim;
  if (m > 0) {
    int rc = rdb_reserve(db, db->n_local + m);
    if (rc) return rc;
    // upload in bounded pieces through the staging buffer
    const int64_t piece = 1 << 16;
    rc = rdb_stage(db, (size_t)(m < piece ? m : piece) * db->dim);
    if (rc) return rc;
    for (int64_t o = 0; o < m; o += piece) {
      const int64_t c = m - o < piece ? m - o : piece;
      DSM_HIP(hipMemcpyAsync(db->d_q, mine.data() + o * db->dim, sizeof(float) * c * db->dim, hipMemcpyHostToDevice,
                             db->ctx->stream));
      launch_ringkey_insert(db->ctx->stream, db->d_keysT, db->cap, db->n_local + o, db->dim, db->d_q, (int)c);
      DSM_HIP(hipStreamSynchronize(db->ctx->stream));
    }
    db->n_local += m;
  }
  db->size_global += n;
  return DSM_OK;
}

static int rdb_knn_dev(dsm_ringdb *db, const float *d_queries, int nq, unsigned long long *d_out) {
  const int n_slices = ringkey_num_slices(db->n_local, nq, db->dim);
  const size_t need = (size_t)n_slices * nq * db->k;
  if (need > db->scratch_words) {
    if (db->d_scratch) DSM_HIP(hipFree(db->d_scratch));
    db->d_scratch = nullptr;
    DSM_HIP(hipMalloc(&db->d_scratch, need * sizeof(unsigned long long)));
    db->scratch_words = need;
  }
  launch_ringkey_knn(db->ctx->stream, db->d_keysT, db->cap, db->n_local, db->dim, db->k, db->thres, db->shard_rank,
                     db->shard_count, d_queries, nq, db->d_scratch, n_slices, d_out);
  DSM_HIP(hipGetLastError());
  return DSM_OK;
}

namespace dsm {
int ringdb_knn_device(dsm_ringdb *db, const float *d_queries, int nq, unsigned long long *d_out) { return rdb_knn_dev(db, d_queries, nq, d_out); }
} // namespace dsm
extern "C" {
int dsm_ringdb_destroy(dsm_ringdb *db);

int dsm_ringdb_create(dsm_context *ctx, int dim, int margin, int k, float thres, const float *dummy_key,
                      int64_t capacity, int shard_rank, int shard_count, dsm_ringdb **out) {
  if (!ctx || !out) return invalid("dsm_ringdb_create: null argument");
  if (dim < 1 || dim > 32) return invalid("dsm_ringdb_create: dim must be in [1,32]");
  if (k < 1 || k > 4) return invalid("dsm_ringdb_create: k must be in [1,4]");
  if (margin < 1) return invalid("dsm_ringdb_create: margin must be >= 1");
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return invalid("dsm_ringdb_create: bad shard");
  DSM_HIP(hipSetDevice(ctx->device));
  dsm_ringdb *db = new dsm_ringdb();
  db->ctx = ctx;
  db->dim = dim;
  db->margin = margin;
  db->k = k;
  db->thres = thres;
  db->shard_rank = shard_rank;
  db->shard_count = shard_count;
  db->queue.assign((size_t)margin * dim, 0.f);
  int rc = rdb_reserve(db, capacity > 16 ? capacity : 16);
  if (rc) {
    dsm_ringdb_destroy(db);
    return rc;
  }
  // index slot 0: the reference's dummy entry (LoopHandler.cpp:35-39, quirk Q8)
  std::vector<float> dummy(dim, 0.f);
  if (dummy_key) memcpy(dummy.data(), dummy_key, sizeof(float) * dim);
  rc = rdb_append(db, dummy.data(), 1);
  if (rc) {
    dsm_ringdb_destroy(db);
    return rc;
  }
  *out = db;
  return DSM_OK;
}

int dsm_ringdb_destroy(dsm_ringdb *db) {
  if (!db) return DSM_OK;
  hipSetDevice(db->ctx->device);
  hipStreamSynchronize(db->ctx->stream);
  hipFree(db->d_keysT);
  hipFree(db->d_q);
  hipFree(db->d_scratch);
  hipFree(db->d_out);
  hipFree(db->d_merge);
  hipFree(db->d_agree);
  dsm::ringdb_forget_comm(db);
  delete db;
  return DSM_OK;
}

int64_t dsm_ringdb_size(dsm_ringdb *db) { return db ? db->size_global : -1; }

int dsm_ringdb_add_points(dsm_ringdb *db, const float *keys, int64_t n_keys) {
  if (!db || !keys || n_keys < 0) return invalid("dsm_ringdb_add_points: bad argument");
  DSM_HIP(hipSetDevice(db->ctx->device));
  return rdb_append(db, keys, n_keys);
}

int dsm_ringdb_enqueue(dsm_ringdb *db, const float *key) { // search_place.h:41-56
  if (!db || !key) return invalid("dsm_ringdb_enqueue: bad argument");
  DSM_HIP(hipSetDevice(db->ctx->device));
  float *slot = db->queue.data() + (size_t)(db->queue_idx % db->margin) * db->dim;
  if (db->queue_idx >= db->margin) {
    int rc = rdb_append(db, slot, 1);
    if (rc) return rc;
  }
  memcpy(slot, key, sizeof(float) * db->dim);
  db->queue_idx++;
  return DSM_OK;
}

int dsm_ringdb_knn_packed_dev(dsm_ringdb *db, const void *d_queries, int nq, void *d_packed_out) {
  if (!db || !d_queries || !d_packed_out || nq < 1) return invalid("dsm_ringdb_knn_packed_dev: bad argument");
  DSM_HIP(hipSetDevice(db->ctx->device));
  return rdb_knn_dev(db, (const float *)d_queries, nq, (unsigned long long *)d_packed_out);
}

int dsm_ringdb_knn_packed(dsm_ringdb *db, const float *queries, int nq, void *d_packed_out) {
  if (!db || !queries || !d_packed_out || nq < 1) return invalid("dsm_ringdb_knn_packed: bad argument");
  DSM_HIP(hipSetDevice(db->ctx->device));
  int rc = rdb_stage(db, (size_t)nq * db->dim);
  if (rc) return rc;
  DSM_HIP(hipMemcpyAsync(db->d_q, queries, sizeof(float) * (size_t)nq * db->dim, hipMemcpyHostToDevice, db->ctx->stream));
  rc = rdb_knn_dev(db, db->d_q, nq, (unsigned long long *)d_packed_out);
  if (rc) return rc;
  DSM_HIP(hipStreamSynchronize(db->ctx->stream));
  return DSM_OK;
}

int dsm_ringdb_knn_packed_host(dsm_ringdb *db, const float *queries, int nq, int64_t *packed_out) {
  if (!db || !queries || !packed_out || nq < 1) return invalid("dsm_ringdb_knn_packed_host: bad argument");
  DSM_HIP(hipSetDevice(db->ctx->device));
  const size_t words = (size_t)nq * db->k;
  if (words > db->out_words) {
    if (db->d_out) DSM_HIP(hipFree(db->d_out));
    db->d_out = nullptr;
    DSM_HIP(hipMalloc(&db->d_out, words * sizeof(unsigned long long)));
    db->out_words = words;
  }
  int rc = dsm_ringdb_knn_packed(db, queries, nq, db->d_out);
  if (rc) return rc;
  DSM_HIP(hipMemcpy(packed_out, db->d_out, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return DSM_OK;
}

int dsm_ringdb_query_then_enqueue(dsm_ringdb *db, const float *key, int *cand_out, int *ncand_out) {
  if (!db || !key || !cand_out || !ncand_out) return invalid("dsm_ringdb_query_then_enqueue: bad argument");
  // Sharded handle: a COLLECTIVE call -- every rank passes the same key, scans its shard, the candidates are merged
  // through the attached communicator (RCCL all-reduce(min)), and every rank returns the same candidate list and
  // enqueues the key (each shard keeps the ordinals that are its own).
  if (db->shard_count != 1 && !db->comm && !db->tr_allreduce) {
    set_error("dsm_ringdb_query_then_enqueue on a sharded DB needs a communicator (dsm_ringdb_attach_comm); "
              "without one use knn_packed + your own merge + enqueue");
    return DSM_ERR_STATE;
  }
  int nc = 0;
  const bool search = db->size_global > db->k; // `ringkeys->size() > FLANN_NN`, search_place.h:29
  int64_t packed[4] = {DSM_RINGDB_NO_CANDIDATE, DSM_RINGDB_NO_CANDIDATE, DSM_RINGDB_NO_CANDIDATE, DSM_RINGDB_NO_CANDIDATE};
  if (db->shard_count == 1) {
    if (search) {
      const int rc = dsm_ringdb_knn_packed_host(db, key, 1, packed);
      if (rc) return rc;
    }
  } else {
    DSM_HIP(hipSetDevice(db->ctx->device));
    // Local part first; whatever goes wrong here is agreed upon by ALL ranks (round 0, also when the index is still too small
    // to be searched: the ranks must agree on that as well) before any of them enters the merge rounds -- a rank that
    // returned early on its own would leave the others waiting in an all-reduce.
    auto local_scan = [&]() -> int {
      if ((size_t)db->k > db->out_words) {
        if (db->d_out) DSM_HIP(hipFree(db->d_out));
        db->d_out = nullptr;
        db->out_words = 0;
        DSM_HIP(hipMalloc(&db->d_out, 4 * sizeof(unsigned long long)));
        db->out_words = 4;
      }
      const int r = rdb_stage(db, (size_t)db->dim);
      if (r) return r;
      DSM_HIP(hipMemcpyAsync(db->d_q, key, sizeof(float) * db->dim, hipMemcpyHostToDevice, db->ctx->stream));
      return rdb_knn_dev(db, db->d_q, 1, db->d_out);
    };
    const int rc_local = search ? local_scan() : DSM_OK;
    const std::string why = rc_local ? dsm_last_error() : "";
    int rc = ringdb_agree(db, rc_local == DSM_OK, why.c_str());
    if (rc) return rc_local ? rc_local : rc;
    if (search) {
      rc = ringdb_merge_attached(db, db->d_out, 1);
      if (rc) return rc;
      DSM_HIP(hipMemcpyAsync(packed, db->d_out, sizeof(unsigned long long) * db->k, hipMemcpyDeviceToHost, db->ctx->stream));
      DSM_HIP(hipStreamSynchronize(db->ctx->stream));
    }
  }
  for (int i = 0; search && i < db->k; i++) {
    if (packed[i] == DSM_RINGDB_NO_CANDIDATE) continue; // dist >= RINGKEY_THRES was filtered on the device
    const int idx = (int)(packed[i] & 0xFFFFFFFFll);
    if (idx > 0) cand_out[nc++] = idx - 1; // :34-38
  }
  *ncand_out = nc;
  return dsm_ringdb_enqueue(db, key);
}

} // extern "C"
