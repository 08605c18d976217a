// The multi-GPU merge of the per-GPU k-mer histograms behind the C-ABI: one process per GPU, RCCL over
// xGMI.  The reference has no multi-device path; what these entry points replace is EncodedCounts.__add__ across the
// chunks of a file (bionumpy/sequence/count_encoded.py:38-55) when the chunks were counted on different GPUs.
//
//   dense histograms (k <= 13)   bnpk_allreduce_hist: ncclAllReduce(int64, sum) over the 4^k bins
//   sparse histograms (k > 13)   the key space is cut into one range per rank; bnpk_exchange_counts tells every rank how
//                                much it will receive, bnpk_exchange_by_key_range moves every item to the rank that owns
//                                its key range in ONE grouped ncclSend/ncclRecv step (each of a GPU's xGMI links carries
//                                the slice of one peer, all of them at once) — raw 8-byte hashes before counting, or
//                                (key, count) runs after a local histogram, whichever moves fewer bytes (parallel.py);
//                                bnpk_exchange_slices is the same step with the slice of every peer at an offset of its
//                                own: one call per group of key ranges, on a stream of its own, while the group before
//                                is counted
//
// RCCL is loaded on first use (dlopen of librccl.so.1 — the copy already in the process if the caller's framework brought
// one — then librccl.so): a library that is only ever used on one GPU does not need it, and bnpk_comm_* report
// BNPK_ERR_NODEVICE instead of failing at load time where it is absent.  `comm` is an ncclComm_t: the caller's own
// communicator, or one made by bnpk_comm_init from an id that rank 0 took with bnpk_comm_unique_id and handed to the
// other ranks by whatever side channel launched them.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct rccl_api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;
thread_local const char* g_last_rccl_error = "";

template <typename F>
bool load_symbol(void* h, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(h, name));
  return out != nullptr;
}

const rccl_api& rccl() {
  std::call_once(g_rccl_once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      rccl_api a;
      a.handle = h;
      const bool all = load_symbol(h, "ncclGetUniqueId", a.GetUniqueId) && load_symbol(h, "ncclCommInitRank", a.CommInitRank) &&
                       load_symbol(h, "ncclCommDestroy", a.CommDestroy) && load_symbol(h, "ncclCommCount", a.CommCount) &&
                       load_symbol(h, "ncclCommUserRank", a.CommUserRank) && load_symbol(h, "ncclAllReduce", a.AllReduce) &&
                       load_symbol(h, "ncclSend", a.Send) && load_symbol(h, "ncclRecv", a.Recv) &&
                       load_symbol(h, "ncclGroupStart", a.GroupStart) && load_symbol(h, "ncclGroupEnd", a.GroupEnd) &&
                       load_symbol(h, "ncclGetErrorString", a.GetErrorString);
      if (all) {
        a.ok = true;
        g_rccl = a;
        return;
      }
      dlclose(h);
    }
  });
  return g_rccl;
}

#define BNPK_RCCL(api, call)                                \
  do {                                                      \
    ncclResult_t r__ = (call);                              \
    if (r__ != ncclSuccess) {                               \
      g_last_rccl_error = (api).GetErrorString(r__);        \
      return BNPK_ERR_HIP;                                  \
    }                                                       \
  } while (0)

// inside ncclGroupStart ... ncclGroupEnd: an error closes the group before it is reported (a group left open swallows
// every later call of the thread)
#define BNPK_RCCL_IN_GROUP(api, call)                       \
  do {                                                      \
    ncclResult_t r__ = (call);                              \
    if (r__ != ncclSuccess) {                               \
      g_last_rccl_error = (api).GetErrorString(r__);        \
      (void)(api).GroupEnd();                               \
      return BNPK_ERR_HIP;                                  \
    }                                                       \
  } while (0)

int comm_shape(const rccl_api& a, ncclComm_t comm, int& world, int& rank) {
  BNPK_RCCL(a, a.CommCount(comm, &world));
  BNPK_RCCL(a, a.CommUserRank(comm, &rank));
  return BNPK_OK;
}

}  // namespace

extern "C" {

const char* bnpk_last_comm_error(void) { return g_last_rccl_error; }

// whether RCCL can be loaded here at all — dlopen + dlsym, nothing made: ncclGetUniqueId would start a bootstrap root (a
// listening socket and a thread) on every rank that merely asks
int bnpk_comm_available(void) { return rccl().ok ? 1 : 0; }

int bnpk_comm_unique_id(uint8_t* id128) {
  if (!id128) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  static_assert(sizeof(ncclUniqueId) == BNPK_COMM_ID_BYTES, "the id is handed around as BNPK_COMM_ID_BYTES bytes");
  ncclUniqueId id;
  BNPK_RCCL(a, a.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return BNPK_OK;
}

int bnpk_comm_init(bnpk_ctx* ctx, const uint8_t* id128, int n_ranks, int rank, void** comm_out) {
  if (!ctx || !id128 || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  BNPK_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  BNPK_RCCL(a, a.CommInitRank(&comm, n_ranks, id, rank));
  *comm_out = comm;
  return BNPK_OK;
}

int bnpk_comm_destroy(void* comm) {
  if (!comm) return BNPK_OK;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  BNPK_RCCL(a, a.CommDestroy((ncclComm_t)comm));
  return BNPK_OK;
}

int bnpk_comm_shape(void* comm, int* n_ranks, int* rank) {
  if (!comm || !n_ranks || !rank) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  return comm_shape(a, (ncclComm_t)comm, *n_ranks, *rank);
}

int bnpk_allreduce_hist(bnpk_ctx* ctx, void* comm, int64_t* d_hist, int64_t bins, void* stream) {
  if (!ctx || !comm || bins < 0 || (bins > 0 && !d_hist)) return BNPK_ERR_ARG;
  if (bins == 0) return BNPK_OK;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  bnpk_timer t(ctx, "allreduce_hist", (hipStream_t)stream);
  BNPK_RCCL(a, a.AllReduce(d_hist, d_hist, (size_t)bins, ncclInt64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return BNPK_OK;
}

int bnpk_exchange_counts(bnpk_ctx* ctx, void* comm, const int64_t* h_send_counts, int n_per_peer, int64_t* h_recv_counts,
                         void* stream) {
  if (!ctx || !comm || !h_send_counts || !h_recv_counts || n_per_peer < 1) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  int world = 0, rank = 0;
  BNPK_CHECK(comm_shape(a, (ncclComm_t)comm, world, rank));
  hipStream_t s = (hipStream_t)stream;
  const size_t words = (size_t)world * n_per_peer;
  void* dev = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 2 * words * 8, &dev, s));
  int64_t* d_send = reinterpret_cast<int64_t*>(dev);
  int64_t* d_recv = d_send + words;
  BNPK_HIP(ctx, hipMemcpyAsync(d_send, h_send_counts, words * 8, hipMemcpyHostToDevice, s));
  BNPK_RCCL(a, a.GroupStart());
  for (int p = 0; p < world; ++p) {
    BNPK_RCCL_IN_GROUP(a, a.Send(d_send + (size_t)p * n_per_peer, (size_t)n_per_peer, ncclInt64, p, (ncclComm_t)comm, s));
    BNPK_RCCL_IN_GROUP(a, a.Recv(d_recv + (size_t)p * n_per_peer, (size_t)n_per_peer, ncclInt64, p, (ncclComm_t)comm, s));
  }
  BNPK_RCCL(a, a.GroupEnd());
  BNPK_HIP(ctx, hipMemcpyAsync(h_recv_counts, d_recv, words * 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  return BNPK_OK;
}

int bnpk_exchange_slices(bnpk_ctx* ctx, void* comm, const int64_t* d_send, const int64_t* h_send_offsets, const int64_t* h_send_counts,
                         int64_t* d_recv, const int64_t* h_recv_counts, void* stream) {
  if (!ctx || !comm || !h_send_offsets || !h_send_counts || !h_recv_counts) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  int world = 0, rank = 0;
  BNPK_CHECK(comm_shape(a, (ncclComm_t)comm, world, rank));
  hipStream_t s = (hipStream_t)stream;
  std::vector<int64_t> recv_off(world + 1, 0);
  int64_t sent = 0;
  for (int p = 0; p < world; ++p) {
    if (h_send_counts[p] < 0 || h_recv_counts[p] < 0 || h_send_offsets[p] < 0) return BNPK_ERR_ARG;
    sent += h_send_counts[p];
    recv_off[p + 1] = recv_off[p] + h_recv_counts[p];
  }
  if ((sent > 0 && !d_send) || (recv_off[world] > 0 && !d_recv)) return BNPK_ERR_ARG;
  if (h_send_counts[rank] != h_recv_counts[rank]) return BNPK_ERR_ARG;
  bnpk_timer t(ctx, "exchange_by_key_range", s);
  // this rank's own slice never leaves the device; every other slice is one send and one receive, all in one group: the
  // transfers to the N - 1 peers run concurrently, one per xGMI link
  if (h_send_counts[rank] > 0)
    BNPK_HIP(ctx, hipMemcpyAsync(d_recv + recv_off[rank], d_send + h_send_offsets[rank], (size_t)h_send_counts[rank] * 8,
                                 hipMemcpyDeviceToDevice, s));
  BNPK_RCCL(a, a.GroupStart());
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    if (h_send_counts[p] > 0)
      BNPK_RCCL_IN_GROUP(a, a.Send(d_send + h_send_offsets[p], (size_t)h_send_counts[p], ncclInt64, p, (ncclComm_t)comm, s));
    if (h_recv_counts[p] > 0)
      BNPK_RCCL_IN_GROUP(a, a.Recv(d_recv + recv_off[p], (size_t)h_recv_counts[p], ncclInt64, p, (ncclComm_t)comm, s));
  }
  BNPK_RCCL(a, a.GroupEnd());
  return BNPK_OK;
}

int bnpk_exchange_by_key_range(bnpk_ctx* ctx, void* comm, const int64_t* d_send, const int64_t* h_send_counts, int64_t* d_recv,
                               const int64_t* h_recv_counts, void* stream) {
  if (!ctx || !comm || !h_send_counts || !h_recv_counts) return BNPK_ERR_ARG;
  const rccl_api& a = rccl();
  if (!a.ok) return BNPK_ERR_NODEVICE;
  int world = 0, rank = 0;
  BNPK_CHECK(comm_shape(a, (ncclComm_t)comm, world, rank));
  std::vector<int64_t> send_off(world, 0);
  for (int p = 1; p < world; ++p) {
    if (h_send_counts[p - 1] < 0) return BNPK_ERR_ARG;
    send_off[p] = send_off[p - 1] + h_send_counts[p - 1];
  }
  return bnpk_exchange_slices(ctx, comm, d_send, send_off.data(), h_send_counts, d_recv, h_recv_counts, stream);
}

}  // extern "C"
