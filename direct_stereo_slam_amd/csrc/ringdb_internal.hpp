// ringdb_internal.hpp -- the ring-key database object behind the opaque dsm_ringdb handle (ringdb_capi.hip, comm_capi.hip)
#pragma once
#include <cstdint>
#include <vector>

#include "dsm_internal.hpp"

struct dsm_ringdb {
  dsm_context *ctx = nullptr;
  int dim = 20, margin = 100, k = 3;
  float thres = 0.1f;
  int shard_rank = 0, shard_count = 1;
  int64_t size_global = 0; // entries in the (global) index, dummy included
  int64_t n_local = 0, cap = 0;
  float *d_keysT = nullptr;
  std::vector<float> queue; // margin x dim ring buffer (search_place.h:43-45)
  int64_t queue_idx = 0;
  float *d_q = nullptr; // query / insert staging
  size_t q_floats = 0;
  unsigned long long *d_scratch = nullptr;
  size_t scratch_words = 0;
  unsigned long long *d_out = nullptr;
  size_t out_words = 0;
  // cross-shard merge (comm_capi.hip): workspace and the communicator attached for query_then_enqueue
  unsigned long long *d_merge = nullptr;
  size_t merge_words = 0;
  struct dsm_comm *comm = nullptr; // borrowed
  // ... or a caller-supplied transport (dsm_ringdb_attach_transport)
  dsm_allreduce_min_u64_fn tr_allreduce = nullptr;
  dsm_allgather_u64_fn tr_allgather = nullptr;
  void *tr_user = nullptr;
  int tr_nranks = 0;
};

namespace dsm {
// merge d_packed (nq x k local candidates) across the shards through the attached communicator
int ringdb_merge_attached(dsm_ringdb *db, void *d_packed, int nq);
} // namespace dsm
