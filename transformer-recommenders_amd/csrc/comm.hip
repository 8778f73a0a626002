// The data-parallel exchange behind the C ABI: SUM all-reduce of the flat gradient buffer over RCCL (xGMI rings).
// Host code only. RCCL is resolved at run time -- dlopen("librccl.so.1"): the copy the process already maps (torch ships one
// with the same SONAME) or the ROCm one -- so libxfmr_hip.so does not link against it and single-GPU users never load it.
// The reference gets this exchange from torch DDP (config.yaml:5-6,35); one message for the whole buffer: 3.1 MiB at the
// MovieLens-1M config is latency-bound, buckets would add latencies.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "internal.h"

namespace {

// the five RCCL entry points used, with RCCL's own types restated (rccl.h is not included: nothing of RCCL is needed to
// BUILD this library): ncclUniqueId is 128 opaque bytes passed BY VALUE, ncclFloat = 7, ncclSum = 0, ncclSuccess = 0
struct UniqueId { char internal[XFMR_COMM_ID_BYTES]; };
using comm_t = void*;
struct Rccl {
  int (*GetUniqueId)(UniqueId*);
  int (*CommInitRank)(comm_t*, int, UniqueId, int);
  int (*CommDestroy)(comm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  bool ok;
};
thread_local char g_err[256] = "";

const Rccl& rccl() {
  static Rccl r = [] {
    Rccl t{};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return t;
    t.GetUniqueId = (decltype(t.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    t.CommInitRank = (decltype(t.CommInitRank))dlsym(h, "ncclCommInitRank");
    t.CommDestroy = (decltype(t.CommDestroy))dlsym(h, "ncclCommDestroy");
    t.AllReduce = (decltype(t.AllReduce))dlsym(h, "ncclAllReduce");
    t.GetErrorString = (decltype(t.GetErrorString))dlsym(h, "ncclGetErrorString");
    t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.AllReduce;
    return t;
  }();
  return r;
}
int fail(const Rccl& r, int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, (r.GetErrorString && code > 0) ? r.GetErrorString(code) : "RCCL not loadable");
  return XFMR_ECOMM;
}

}  // namespace

extern "C" {

const char* xfmr_comm_last_error(void) { return g_err; }

int xfmr_comm_unique_id(unsigned char id[XFMR_COMM_ID_BYTES]) {
  if (!id) return XFMR_EINVAL;
  const Rccl& r = rccl();
  if (!r.ok) return fail(r, 0, "xfmr_comm_unique_id");
  UniqueId u;
  if (int rc = r.GetUniqueId(&u)) return fail(r, rc, "ncclGetUniqueId");
  memcpy(id, u.internal, XFMR_COMM_ID_BYTES);
  return XFMR_OK;
}

int xfmr_comm_create(void** comm, const unsigned char id[XFMR_COMM_ID_BYTES], int32_t world, int32_t rank) {
  if (!comm || !id || world < 1 || rank < 0 || rank >= world) return XFMR_EINVAL;
  const Rccl& r = rccl();
  if (!r.ok) return fail(r, 0, "xfmr_comm_create");
  UniqueId u;
  memcpy(u.internal, id, XFMR_COMM_ID_BYTES);
  comm_t c = nullptr;
  if (int rc = r.CommInitRank(&c, world, u, rank)) return fail(r, rc, "ncclCommInitRank");
  *comm = c;
  return XFMR_OK;
}

int xfmr_comm_destroy(void* comm) {
  if (!comm) return XFMR_EINVAL;
  const Rccl& r = rccl();
  if (!r.ok) return fail(r, 0, "xfmr_comm_destroy");
  if (int rc = r.CommDestroy((comm_t)comm)) return fail(r, rc, "ncclCommDestroy");
  return XFMR_OK;
}

int xfmr_allreduce_flat(void* comm, float* grads, int64_t n, void* stream) {
  if (!comm || !grads || n <= 0) return XFMR_EINVAL;
  const Rccl& r = rccl();
  if (!r.ok) return fail(r, 0, "xfmr_allreduce_flat");
  if (int rc = r.AllReduce(grads, grads, (size_t)n, /*ncclFloat*/ 7, /*ncclSum*/ 0, (comm_t)comm, (hipStream_t)stream))
    return fail(r, rc, "ncclAllReduce");
  return XFMR_OK;
}

}  // extern "C"
