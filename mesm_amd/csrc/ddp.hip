// Data-parallel gradient exchange over RCCL, owned by this library (ddp_{init, allreduce_bucket, wait};
// the reference is single-process, the insertion point is train.py:68-72).  torch.distributed stays the rendezvous
// (its store carries the 128-byte RCCL unique id), but the collectives of the hot path are issued HERE, on HIP streams
// this library picks, so that they can be recorded into the step's HIP graph: torch's process group has a watchdog
// thread that polls the events of the collectives it issued and aborts the process when such an event belongs to a
// capturing stream (round 2: ~3 % of process starts).  A raw ncclAllReduce has no watcher.
//
// RCCL is bound at run time (dlopen of the librccl.so.1 the process already has -- PyTorch ships one -- or the
// system's): no link-time dependency, and never two copies of the library in one process.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include "common.hpp"

namespace {

typedef struct { char internal[128]; } RcclId;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(RcclComm*, int, RcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
typedef int (*fn_destroy)(RcclComm);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(RcclComm, int*);
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;  // ncclFloat32, ncclSum (rccl.h)

struct Api {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
  fn_count count = nullptr;
};
Api g_api;
char g_err[256] = "";

void set_err(const char* what, int code) {
  const char* s = (g_api.errstr && code > 0) ? g_api.errstr(code) : "";
  snprintf(g_err, sizeof g_err, "%s (code %d) %s", what, code, s);
}

bool load_api() {
  if (g_api.lib) return true;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process, if any
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { snprintf(g_err, sizeof g_err, "librccl.so.1 not found: %s", dlerror()); return false; }
  Api a;
  a.lib = h;
  a.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  a.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  a.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  a.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  a.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  a.count = (fn_count)dlsym(h, "ncclCommCount");
  if (!a.get_id || !a.init_rank || !a.allreduce || !a.destroy) {
    snprintf(g_err, sizeof g_err, "librccl.so.1 lacks a required symbol");
    return false;
  }
  g_api = a;
  return true;
}

struct Ddp {
  RcclComm comm = nullptr;
  hipStream_t side = nullptr;  // the communicator's own stream (overlapped mode)
  hipEvent_t ev = nullptr;
  int world = 1;
  bool side_used = false;
};

}  // namespace

extern "C" const char* mesm_ddp_last_error(void) { return g_err; }

extern "C" int mesm_ddp_unique_id(uint8_t* out128) {
  if (!out128) return MESM_EINVAL;
  if (!load_api()) return MESM_ELAUNCH;
  RcclId id;
  const int rc = g_api.get_id(&id);
  if (rc != 0) { set_err("ncclGetUniqueId", rc); return MESM_ELAUNCH; }
  memcpy(out128, id.internal, 128);
  return MESM_OK;
}

extern "C" int mesm_ddp_init(const uint8_t* id128, int32_t rank, int32_t world, void** handle) {
  if (!id128 || !handle || world < 1 || rank < 0 || rank >= world) return MESM_EINVAL;
  if (!load_api()) return MESM_ELAUNCH;
  Ddp* d = new Ddp;
  d->world = world;
  RcclId id;
  memcpy(id.internal, id128, 128);
  int rc = g_api.init_rank(&d->comm, world, id, rank);
  if (rc != 0) { set_err("ncclCommInitRank", rc); delete d; return MESM_ELAUNCH; }
  if (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&d->ev, hipEventDisableTiming) != hipSuccess) {
    set_err("hipStreamCreate / hipEventCreate", -1);
    g_api.destroy(d->comm);
    delete d;
    return MESM_ELAUNCH;
  }
  *handle = d;
  return MESM_OK;
}

// In-place sum over the ranks of buf[0, count).  side = 0: on `stream` itself (the step stays one linear chain of its
// queue; the wire time is exposed).  side = 1: on the communicator's own stream, ordered behind everything enqueued
// on `stream` so far (event fork) -- the bucket reduces while `stream` goes on with backward; mesm_ddp_wait joins.
// Capturable either way: under stream capture the fork pulls the side stream into the capture.
extern "C" int mesm_ddp_allreduce(void* handle, float* buf, int64_t count, void* stream, int32_t side) {
  Ddp* d = (Ddp*)handle;
  if (!d || !buf || count <= 0) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipStream_t on = s;
  if (side) {
    if (hipEventRecord(d->ev, s) != hipSuccess || hipStreamWaitEvent(d->side, d->ev, 0) != hipSuccess) {
      set_err("fork to the collective stream", -1);
      return MESM_ELAUNCH;
    }
    on = d->side;
    d->side_used = true;
  }
  const int rc = g_api.allreduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, d->comm, on);
  if (rc != 0) { set_err("ncclAllReduce", rc); return MESM_ELAUNCH; }
  return MESM_OK;
}

// `stream` waits for every collective issued on the communicator's own stream so far (no-op if none was).
extern "C" int mesm_ddp_wait(void* handle, void* stream) {
  Ddp* d = (Ddp*)handle;
  if (!d) return MESM_EINVAL;
  if (!d->side_used) return MESM_OK;
  if (hipEventRecord(d->ev, d->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, d->ev, 0) != hipSuccess) {
    set_err("join of the collective stream", -1);
    return MESM_ELAUNCH;
  }
  d->side_used = false;
  return MESM_OK;
}

// Ranks the communicator actually spans, as RCCL itself reports it (ncclCommCount): what a first multi-GPU run prints
// to show that the library's own communicator, not a 1-rank stand-in, carried the gradients.
extern "C" int mesm_ddp_count(void* handle, int32_t* ranks) {
  Ddp* d = (Ddp*)handle;
  if (!d || !ranks) return MESM_EINVAL;
  if (!g_api.count) { *ranks = d->world; return MESM_OK; }
  int n = 0;
  const int rc = g_api.count(d->comm, &n);
  if (rc != 0) { set_err("ncclCommCount", rc); return MESM_ELAUNCH; }
  *ranks = n;
  return MESM_OK;
}

extern "C" int mesm_ddp_destroy(void* handle) {
  Ddp* d = (Ddp*)handle;
  if (!d) return MESM_EINVAL;
  if (d->comm) g_api.destroy(d->comm);
  if (d->ev) hipEventDestroy(d->ev);
  if (d->side) hipStreamDestroy(d->side);
  delete d;
  return MESM_OK;
}
