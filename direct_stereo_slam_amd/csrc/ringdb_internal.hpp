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
  struct dsm_comm *comm = nullptr; // borrowed; dsm_comm_destroy detaches it from every database it is attached to
  unsigned long long *d_agree = nullptr; // 4 words: the ranks' agreement round of a collective query (allocated when a
                                         // communicator / transport is attached, so that the query path allocates nothing)
  // ... or a caller-supplied transport (dsm_ringdb_attach_transport)
  dsm_allreduce_min_u64_fn tr_allreduce = nullptr;
  dsm_allgather_u64_fn tr_allgather = nullptr;
  void *tr_user = nullptr;
  int tr_nranks = 0;
};

namespace dsm {
// merge d_packed (nq x k local candidates) across the shards through the attached communicator
int ringdb_merge_attached(dsm_ringdb *db, void *d_packed, int nq);
// Round 0 of a collective query: one all-reduce(min) over {this rank is ready, index size, 2^62 - index size}.  Returns DSM_OK on
// every rank only if every rank was ready and all hold the same index size; otherwise every rank gets the same error and
// none enters the merge rounds (a rank that bailed out alone would leave the others waiting in a collective).
int ringdb_agree(dsm_ringdb *db, bool ready, const char *why_not);
// the local k-NN scan over device-resident queries (nq x dim floats) into d_out (nq x k packed candidates), enqueued on the context's stream
int ringdb_knn_device(dsm_ringdb *db, const float *d_queries, int nq, unsigned long long *d_out);
// the communicator is going away: detach it from the databases that borrow it
void ringdb_forget_comm(dsm_ringdb *db);
} // namespace dsm
