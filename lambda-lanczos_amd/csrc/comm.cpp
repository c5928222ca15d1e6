// RCCL communicator for the 1-D row-partitioned multi-GPU mode: one process per GPU, collectives
// enqueued on the context's HIP stream so that they order with the kernels without host synchronisation.
//
// librccl is loaded lazily with dlopen: single-GPU users need no RCCL at all, and when the host process has
// already loaded an RCCL (e.g. PyTorch-ROCm bundles one under the same SONAME librccl.so.1) that very copy is
// reused instead of pulling a second one into the process.
//
// Exchange steps per Lanczos iteration: one all-gather of the current Lanczos vector shard (so every rank holds
// the full x for its row block) and a few all-reduce(sum) of k+1 doubles for the block Gram-Schmidt coefficients,
// alpha and the norms.  xGMI is point to point, so the all-gather is the per-link-bound step; everything else is
// latency sized.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "ll_internal.hpp"

namespace ll {

namespace {
struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Api& api() {
  static Api a;
  if (a.lib) return a;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* nm : names) {  // prefer a copy that is already mapped into the process
    a.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (a.lib) break;
  }
  if (!a.lib)
    for (const char* nm : names) {
      a.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
  if (!a.lib) {
    set_error(std::string("cannot load librccl: ") + dlerror());
    throw Failure{LL_ERR_RCCL};
  }
  auto sym = [&](const char* s) {
    void* p = dlsym(a.lib, s);
    if (!p) {
      set_error(std::string("librccl lacks symbol ") + s);
      throw Failure{LL_ERR_RCCL};
    }
    return p;
  };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
  a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  return a;
}

void check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) {
    set_error(std::string(what) + " failed: " + api().GetErrorString(r));
    throw Failure{LL_ERR_RCCL};
  }
}
}  // namespace

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
};

void comm_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == LL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId id;
  check(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof(id));
}

Comm* comm_create(const void* id128, int rank, int nranks, int device) {
  LL_HIP(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  Comm* c = new Comm;
  c->rank = rank;
  c->nranks = nranks;
  try {
    check(api().CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");
  } catch (...) {
    delete c;
    throw;
  }
  return c;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm) api().CommDestroy(c->comm);
  delete c;
}

void comm_allgather(Comm* c, const void* send, void* recv, size_t n_doubles, hipStream_t s) {
  check(api().AllGather(send, recv, n_doubles, ncclDouble, c->comm, s), "ncclAllGather");
}

void comm_allreduce_sum(Comm* c, double* buf, size_t n_doubles, hipStream_t s) {
  check(api().AllReduce(buf, buf, n_doubles, ncclDouble, ncclSum, c->comm, s), "ncclAllReduce");
}

}  // namespace ll
