// Collective plumbing for hosts that do not bring their own: the three exchange points of the data-parallel path
// (SwitchNorm batch statistics, gradients, confusion matrix) as in-place all-reduces over RCCL.  Replaces the
// reference's thread rendez-vous of modeling/sync_batchnorm/comm.py:39-43,71-90 (SyncMaster / SlavePipe) and
// DataParallel's gradient reduction (train.py:85,253).  RCCL is loaded at run time (dlopen), so the library has no
// link-time dependency on it and a host that already runs collectives (torch.distributed) never touches this file.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef int (*fn_get_unique_id)(void*);
struct Id128 { char b[128]; };   // ncclUniqueId is passed BY VALUE to ncclCommInitRank
typedef int (*fn_init)(void**, int, Id128, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

struct Api {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_init init = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
};

Api g_api;

int load_api() {
  if (g_api.handle) return SNWS_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    snws_set_error("snws_comm: cannot load librccl (%s)", dlerror());
    return SNWS_ERR_HIP;
  }
  Api a;
  a.handle = h;
  a.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  a.init = (fn_init)dlsym(h, "ncclCommInitRank");
  a.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  a.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  a.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!a.get_unique_id || !a.init || !a.allreduce || !a.destroy) {
    snws_set_error("snws_comm: librccl lacks an expected symbol");
    dlclose(h);
    return SNWS_ERR_HIP;
  }
  g_api = a;
  return SNWS_OK;
}

int check(int rc, const char* what) {
  if (rc == 0) return SNWS_OK;
  snws_set_error("snws_comm: %s failed: %s", what, g_api.errstr ? g_api.errstr(rc) : "rccl error");
  return SNWS_ERR_HIP;
}

// ncclDataType_t / ncclRedOp_t values of rccl.h
constexpr int NCCL_INT64 = 4, NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_MAX = 2;

}  // namespace

struct snws_comm {
  void* nccl;
  int rank, world;
};

extern "C" int snws_comm_unique_id(void* id128) {
  SNWS_CHECK_ARG(id128, "snws_comm_unique_id: NULL buffer");
  if (int rc = load_api()) return rc;
  return check(g_api.get_unique_id(id128), "ncclGetUniqueId");
}

extern "C" int snws_comm_init(const void* id128, int rank, int world, snws_comm** out) {
  SNWS_CHECK_ARG(id128 && out, "snws_comm_init: NULL pointer");
  SNWS_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "snws_comm_init: rank %d outside world %d", rank, world);
  if (int rc = load_api()) return rc;
  Id128 id;
  memcpy(id.b, id128, sizeof(id.b));
  void* c = nullptr;
  if (int rc = check(g_api.init(&c, world, id, rank), "ncclCommInitRank")) return rc;
  snws_comm* h = new snws_comm{c, rank, world};
  *out = h;
  return SNWS_OK;
}

extern "C" int snws_comm_allreduce_f32(snws_comm* comm, float* buf, int64_t count, void* stream) {
  SNWS_CHECK_ARG(comm && (buf || count == 0) && count >= 0, "snws_comm_allreduce_f32: bad arguments");
  if (count == 0) return SNWS_OK;
  return check(g_api.allreduce(buf, buf, (size_t)count, NCCL_FLOAT32, NCCL_SUM, comm->nccl, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int snws_comm_allreduce_i64(snws_comm* comm, int64_t* buf, int64_t count, int op, void* stream) {
  SNWS_CHECK_ARG(comm && (buf || count == 0) && count >= 0, "snws_comm_allreduce_i64: bad arguments");
  SNWS_CHECK_ARG(op == 0 || op == 1, "snws_comm_allreduce_i64: op %d is neither 0 (sum) nor 1 (max)", op);
  if (count == 0) return SNWS_OK;
  return check(g_api.allreduce(buf, buf, (size_t)count, NCCL_INT64, op == 0 ? NCCL_SUM : NCCL_MAX, comm->nccl, (hipStream_t)stream),
               "ncclAllReduce");
}

extern "C" int snws_comm_destroy(snws_comm* comm) {
  if (!comm) return SNWS_OK;
  int rc = g_api.destroy ? check(g_api.destroy(comm->nccl), "ncclCommDestroy") : SNWS_OK;
  delete comm;
  return rc;
}
