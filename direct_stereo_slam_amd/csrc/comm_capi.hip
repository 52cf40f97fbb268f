// comm_capi.hip -- cross-GPU merge of the sharded ring-key database behind the C ABI.
//
// One process per GPU; each holds the shard `ordinal mod G` of the index that LoopHandler.cpp:35-39 builds and
// search_ringkey (search_place.h:25-39) queries.  Every rank scans its shard (ringkey_kernels.hip) into k packed
// candidates per query -- (float_bits(dist2) << 32) | global index, ascending, so unsigned 64-bit min means "nearest,
// then smallest index" -- and the global top-k is formed by
//   * DSM_MERGE_ALLREDUCE_MIN: k rounds of ncclAllReduce(ncclMin, ncclUint64) over the ranks' current heads; the rank
//     whose head won pops it (candidates are unique, so exactly one rank pops).  A slot-wise min of the sorted
//     triples would NOT be a top-k; the rounds are.  8*Q bytes per round: pure latency on xGMI.
//   * DSM_MERGE_ALLGATHER: one ncclAllGather of the k*Q candidates of every rank + a local k-way merge: one collective.
// Both give the same (bit-identical) result.  The collective is pluggable (dsm_ringdb_merge_topk_with) so that hosts with
// another transport -- and the tests, which run G "ranks" as threads or gloo processes on one GPU -- use the same merge
// kernels; dsm_ringdb_merge_topk binds it to RCCL, loaded at run time from librccl.so.1 (no link-time dependency: a
// single-GPU user never loads it).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <algorithm>
#include <mutex>
#include <vector>

#include "dsm_internal.hpp"
#include "ringdb_internal.hpp"

using namespace dsm;

namespace {

int invalid(const char *m) {
  set_error(m);
  return DSM_ERR_INVALID;
}

// ---- RCCL, resolved at run time ----------------------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

int load_rccl() {
  std::call_once(g_rccl_once, [] {
    // a process that already mapped RCCL (e.g. through torch.distributed) keeps using that copy: same SONAME
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char *n : names) {
      g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) return;
#define DSM_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name))
    DSM_SYM(GetUniqueId, "ncclGetUniqueId");
    DSM_SYM(CommInitRank, "ncclCommInitRank");
    DSM_SYM(CommDestroy, "ncclCommDestroy");
    DSM_SYM(AllReduce, "ncclAllReduce");
    DSM_SYM(AllGather, "ncclAllGather");
    DSM_SYM(GetErrorString, "ncclGetErrorString");
#undef DSM_SYM
  });
  if (!g_rccl.handle || !g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.AllGather) {
    set_error("librccl.so.1 could not be loaded (needed only for the multi-GPU ring-key merge)");
    return DSM_ERR_STATE;
  }
  return DSM_OK;
}

int rccl_fail(ncclResult_t r, const char *what) {
  set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
  return DSM_ERR_HIP;
}
#define DSM_RCCL(expr)                                   \
  do {                                                   \
    ncclResult_t r__ = (expr);                           \
    if (r__ != ncclSuccess) return rccl_fail(r__, #expr); \
  } while (0)

// ---- merge kernels ------------------------------------------------------------------------------------------------
constexpr unsigned long long kNoCand = 0x7FFFFFFFFFFFFFFFull;

// round r of the all-reduce(min) form, fused: consume the winners of round r-1 (write them to out, pop if ours), then
// publish this rank's next head.  win == nullptr: first round.
// (win and head are the same buffer: the all-reduce runs in place)
__global__ void merge_round_kernel(const unsigned long long *__restrict__ local, int *__restrict__ cursor,
                                   const unsigned long long *win, unsigned long long *__restrict__ out, int round,
                                   unsigned long long *head, int nq, int k) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  int c = round == 0 ? 0 : cursor[q];
  if (round > 0) {
    const unsigned long long w = win[q];
    out[(size_t)q * k + (round - 1)] = w;
    if (w != kNoCand && c < k && local[(size_t)q * k + c] == w) c++;
  }
  cursor[q] = c;
  if (head) head[q] = c < k ? local[(size_t)q * k + c] : kNoCand;
}

// all-gather form: all[g][q][k] sorted lists of the G ranks -> the k smallest of their union, ascending
__global__ void merge_gathered_kernel(const unsigned long long *__restrict__ all, int G, int nq, int k,
                                      unsigned long long *__restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  unsigned long long best[4] = {kNoCand, kNoCand, kNoCand, kNoCand};
  for (int g = 0; g < G; g++)
    for (int j = 0; j < k; j++) {
      unsigned long long c = all[((size_t)g * nq + q) * k + j];
#pragma unroll
      for (int s = 0; s < 4; s++) { // insertion into the sorted quadruple; candidates of different ranks are distinct
        const bool lt = c < best[s];
        const unsigned long long lo = lt ? c : best[s];
        c = lt ? best[s] : c;
        best[s] = lo;
      }
    }
  for (int j = 0; j < k; j++) out[(size_t)q * k + j] = best[j];
}

} // namespace

struct dsm_comm {
  dsm_context *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  std::vector<dsm_ringdb *> attached; // databases that borrow this communicator (detached by dsm_comm_destroy)
};

namespace {

// workspace of a merge on the ring DB handle (grown on demand)
int merge_workspace(dsm_ringdb *db, size_t words) {
  if (words <= db->merge_words) return DSM_OK;
  if (db->d_merge) DSM_HIP(hipFree(db->d_merge));
  db->d_merge = nullptr;
  db->merge_words = 0;
  DSM_HIP(hipMalloc(&db->d_merge, words * sizeof(unsigned long long)));
  db->merge_words = words;
  return DSM_OK;
}

int rccl_allreduce_min(void *user, void *d_buf, size_t count, void *stream) {
  dsm_comm *c = (dsm_comm *)user;
  DSM_RCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclUint64, ncclMin, c->comm, (hipStream_t)stream));
  return DSM_OK;
}
int rccl_allgather(void *user, const void *d_send, void *d_recv, size_t count, void *stream) {
  dsm_comm *c = (dsm_comm *)user;
  DSM_RCCL(g_rccl.AllGather(d_send, d_recv, count, ncclUint64, c->comm, (hipStream_t)stream));
  return DSM_OK;
}

} // namespace

extern "C" {

int dsm_comm_unique_id(unsigned char id_out[DSM_COMM_ID_BYTES]) {
  if (!id_out) return invalid("dsm_comm_unique_id: null argument");
  static_assert(DSM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is an ncclUniqueId");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  DSM_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, DSM_COMM_ID_BYTES);
  return DSM_OK;
}

int dsm_comm_create(dsm_context *ctx, const unsigned char id[DSM_COMM_ID_BYTES], int rank, int nranks, dsm_comm **out) {
  if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return invalid("dsm_comm_create: bad argument");
  int rc = load_rccl();
  if (rc) return rc;
  DSM_HIP(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(uid.internal, id, DSM_COMM_ID_BYTES);
  dsm_comm *c = new dsm_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->nranks = nranks;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail(r, "ncclCommInitRank");
  }
  *out = c;
  return DSM_OK;
}

// Destroy order: databases may outlive the communicator (they are detached here: a later collective query fails with
// DSM_ERR_STATE instead of touching freed memory); the communicator must be destroyed BEFORE its context.
int dsm_comm_destroy(dsm_comm *c) {
  if (!c) return DSM_OK;
  for (dsm_ringdb *db : c->attached)
    if (db->comm == c) db->comm = nullptr;
  c->attached.clear();
  hipSetDevice(c->ctx->device);
  hipStreamSynchronize(c->ctx->stream);
  if (c->comm) g_rccl.CommDestroy(c->comm);
  delete c;
  return DSM_OK;
}

int dsm_comm_rank(dsm_comm *c) { return c ? c->rank : -1; }
int dsm_comm_size(dsm_comm *c) { return c ? c->nranks : -1; }

int dsm_ringdb_merge_topk_with(dsm_ringdb *db, void *d_packed, int nq, int algo, int nranks, dsm_allreduce_min_u64_fn allreduce_min,
                               dsm_allgather_u64_fn allgather, void *user) {
  if (!db || !d_packed || nq < 1 || nranks < 1) return invalid("dsm_ringdb_merge_topk: bad argument");
  if (algo == DSM_MERGE_ALLREDUCE_MIN ? !allreduce_min : (algo != DSM_MERGE_ALLGATHER || !allgather))
    return invalid("dsm_ringdb_merge_topk: no collective for this algorithm");
  dsm_context *ctx = db->ctx;
  DSM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int k = db->k;
  unsigned long long *local = (unsigned long long *)d_packed;
  const dim3 grid((nq + 255) / 256), block(256);
  if (algo == DSM_MERGE_ALLREDUCE_MIN) {
    // [head nq][out nq*k][cursor nq (ints, in the space of nq words)]
    int rc = merge_workspace(db, (size_t)nq * (k + 2));
    if (rc) return rc;
    unsigned long long *head = db->d_merge, *outb = head + nq;
    int *cursor = (int *)(outb + (size_t)nq * k);
    for (int r = 0; r <= k; r++) {
      hipLaunchKernelGGL(merge_round_kernel, grid, block, 0, st, local, cursor, r == 0 ? nullptr : head, outb, r, r < k ? head : nullptr, nq, k);
      if (r < k) {
        rc = allreduce_min(user, head, (size_t)nq, (void *)st);
        if (rc) return rc;
      }
    }
    DSM_HIP(hipGetLastError());
    DSM_HIP(hipMemcpyAsync(local, outb, sizeof(unsigned long long) * (size_t)nq * k, hipMemcpyDeviceToDevice, st));
  } else {
    int rc = merge_workspace(db, (size_t)nq * k * nranks);
    if (rc) return rc;
    rc = allgather(user, local, db->d_merge, (size_t)nq * k, (void *)st);
    if (rc) return rc;
    hipLaunchKernelGGL(merge_gathered_kernel, grid, block, 0, st, db->d_merge, nranks, nq, k, local);
    DSM_HIP(hipGetLastError());
  }
  return DSM_OK;
}

int dsm_ringdb_merge_topk(dsm_ringdb *db, dsm_comm *comm, void *d_packed, int nq, int algo) {
  if (!db || !comm) return invalid("dsm_ringdb_merge_topk: null argument");
  if (comm->ctx != db->ctx) return invalid("dsm_ringdb_merge_topk: communicator and database belong to different contexts");
  if (comm->nranks != db->shard_count || comm->rank != db->shard_rank)
    return invalid("dsm_ringdb_merge_topk: the communicator's rank / size must equal the database's shard rank / count");
  return dsm_ringdb_merge_topk_with(db, d_packed, nq, algo, comm->nranks, rccl_allreduce_min, rccl_allgather, comm);
}

int dsm_ringdb_attach_comm(dsm_ringdb *db, dsm_comm *comm) {
  if (!db) return invalid("dsm_ringdb_attach_comm: null database");
  if (comm && (comm->ctx != db->ctx || comm->nranks != db->shard_count || comm->rank != db->shard_rank))
    return invalid("dsm_ringdb_attach_comm: the communicator's context / rank / size must match the database's shard");
  if (db->comm && db->comm != comm) ringdb_forget_comm(db);
  if (comm && !db->d_agree) {
    DSM_HIP(hipSetDevice(db->ctx->device)); // (a process that drives several devices: the buffer belongs on the context's)
    DSM_HIP(hipMalloc(&db->d_agree, 4 * sizeof(unsigned long long)));
  }
  db->comm = comm;
  if (comm && std::find(comm->attached.begin(), comm->attached.end(), db) == comm->attached.end()) comm->attached.push_back(db);
  return DSM_OK;
}

int dsm_ringdb_attach_transport(dsm_ringdb *db, int nranks, dsm_allreduce_min_u64_fn allreduce_min, dsm_allgather_u64_fn allgather, void *user) {
  if (!db) return invalid("dsm_ringdb_attach_transport: null database");
  if (allreduce_min && nranks != db->shard_count) return invalid("dsm_ringdb_attach_transport: nranks must equal the database's shard count");
  if (allreduce_min && !db->d_agree) {
    DSM_HIP(hipSetDevice(db->ctx->device));
    DSM_HIP(hipMalloc(&db->d_agree, 4 * sizeof(unsigned long long)));
  }
  db->tr_allreduce = allreduce_min;
  db->tr_allgather = allgather;
  db->tr_user = user;
  db->tr_nranks = allreduce_min ? nranks : 0;
  return DSM_OK;
}

} // extern "C"

// used by dsm_ringdb_query_then_enqueue on sharded handles (ringdb_capi.hip)
int dsm::ringdb_merge_attached(dsm_ringdb *db, void *d_packed, int nq) {
  if (!db->comm && db->tr_allreduce)
    return dsm_ringdb_merge_topk_with(db, d_packed, nq, DSM_MERGE_ALLREDUCE_MIN, db->tr_nranks, db->tr_allreduce, db->tr_allgather, db->tr_user);
  if (!db->comm) {
    set_error("sharded ring-key DB: attach a communicator first (dsm_ringdb_attach_comm) -- every rank then calls "
              "dsm_ringdb_query_then_enqueue collectively with the same key");
    return DSM_ERR_STATE;
  }
  return dsm_ringdb_merge_topk(db, db->comm, d_packed, nq, DSM_MERGE_ALLREDUCE_MIN);
}

void dsm::ringdb_forget_comm(dsm_ringdb *db) {
  if (!db->comm) return;
  auto &v = db->comm->attached;
  v.erase(std::remove(v.begin(), v.end(), db), v.end());
  db->comm = nullptr;
}

int dsm::ringdb_agree(dsm_ringdb *db, bool ready, const char *why_not) {
  if (!db->comm && !db->tr_allreduce) { // no transport at all: nobody is waiting for this rank in a collective
    set_error("sharded ring-key DB: attach a communicator first (dsm_ringdb_attach_comm)");
    return DSM_ERR_STATE;
  }
  hipStream_t st = db->ctx->stream;
  // A failure on THIS rank before the collective must not make it leave alone -- the other ranks would wait in the
  // all-reduce for ever, which is what this round exists to prevent: it becomes "not ready" and the rank still takes part.
  std::string local_err;
  if (!db->d_agree) { // (attached without the buffer: make it now)
    hipError_t e = hipSetDevice(db->ctx->device);
    if (e == hipSuccess) e = hipMalloc(&db->d_agree, 4 * sizeof(unsigned long long));
    if (e != hipSuccess) {
      hip_fail(e, "ringdb_agree: allocating the agreement words", __FILE__, __LINE__);
      return DSM_ERR_HIP; // without a device buffer this rank cannot enter the collective at all
    }
  }
  // (every word stays below 2^63: transports may order the words as signed integers, as the packed candidates allow)
  const unsigned long long kBig = 1ull << 62, sz = (unsigned long long)db->size_global;
  unsigned long long w[3] = {ready ? 1ull : 0ull, sz, kBig - sz};
  {
    hipError_t e = hipMemcpyAsync(db->d_agree, w, sizeof w, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { // say "not ready" by other means and go on into the collective
      local_err = std::string("staging the agreement words failed: ") + hipGetErrorString(e);
      ready = false;
      (void)hipMemsetAsync(db->d_agree, 0, sizeof w, st);
    }
  }
  int rc = db->comm ? rccl_allreduce_min(db->comm, db->d_agree, 3, (void *)st) : db->tr_allreduce(db->tr_user, db->d_agree, 3, (void *)st);
  if (rc) return rc;
  if (!local_err.empty()) {
    (void)hipStreamSynchronize(st);
    set_error("collective ring-key query: this rank could not take part: " + local_err);
    return DSM_ERR_STATE;
  }
  DSM_HIP(hipMemcpyAsync(w, db->d_agree, sizeof w, hipMemcpyDeviceToHost, st));
  DSM_HIP(hipStreamSynchronize(st));
  if (w[0] != 1ull) {
    set_error(ready ? "collective ring-key query: another rank could not take part (see its dsm_last_error); no rank entered the merge"
                    : (std::string("collective ring-key query: this rank could not take part: ") + (why_not ? why_not : "")).c_str());
    return DSM_ERR_STATE;
  }
  if (w[1] != kBig - w[2]) { // min of the sizes != max of the sizes
    set_error("collective ring-key query: the ranks' databases hold different numbers of entries (every rank must make the same "
              "add_points / query_then_enqueue calls); no rank entered the merge");
    return DSM_ERR_STATE;
  }
  return DSM_OK;
}
