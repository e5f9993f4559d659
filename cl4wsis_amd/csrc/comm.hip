// Data-parallel communication behind the C ABI: one RCCL communicator per process, gradient buckets all-reduced in place
// on a caller-supplied HIP stream. Replaces init_process_group('nccl') + the three DistributedDataParallel wrappers of the
// reference (run.py:38, train.py:187-196) on the data path.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a single-GPU process never loads it, and inside a PyTorch process
// the already-mapped copy (torch/lib/librccl.so has the same SONAME) is the one that is used, so there is exactly one RCCL
// in the address space. Only the handful of entry points below are needed; their prototypes are restated from
// /opt/rocm/include/rccl/rccl.h (RCCL 2.x ABI: ncclUniqueId = 128 opaque bytes passed by value).
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_h;
enum { NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_AVG = 4 };

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
  int (*CommInitRank)(ncclComm_h*, int, ncclUniqueId_t, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_h, hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_h) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
ncclComm_h g_comm = nullptr;
int g_rank = -1, g_world = 0;

int bind_rccl() {
  if (g_rccl.lib) return CL4_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { cl4::set_error("comm: cannot load librccl.so.1 (%s)", dlerror()); return CL4_EINVAL; }
  Rccl r;
  r.lib = lib;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
    cl4::set_error("comm: librccl is missing an entry point");
    return CL4_EINVAL;
  }
  g_rccl = r;
  return CL4_OK;
}

int check(int rc, const char* what) {
  if (rc == 0) return CL4_OK;
  cl4::set_error("comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return CL4_EINVAL;
}

}  // namespace

extern "C" int cl4_comm_unique_id(void* id128) {
  CL4_REQUIRE(id128, "comm_unique_id: null pointer");
  int rc = bind_rccl();
  if (rc) return rc;
  ncclUniqueId_t id;
  rc = check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(id128, &id, sizeof(id));
  return CL4_OK;
}

extern "C" int cl4_comm_init(int rank, int world, const void* unique_id128) {
  CL4_REQUIRE(unique_id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
  CL4_REQUIRE(g_comm == nullptr, "comm_init: a communicator already exists in this process (one per process)");
  int rc = bind_rccl();
  if (rc) return rc;
  ncclUniqueId_t id;
  memcpy(&id, unique_id128, sizeof(id));
  ncclComm_h c = nullptr;
  rc = check(g_rccl.CommInitRank(&c, world, id, rank), "ncclCommInitRank");   // uses the calling thread's current device
  if (rc) return rc;
  g_comm = c;
  g_rank = rank;
  g_world = world;
  return CL4_OK;
}

extern "C" int cl4_comm_world(void) { return g_comm ? g_world : 0; }
extern "C" int cl4_comm_rank(void) { return g_comm ? g_rank : -1; }

extern "C" int cl4_allreduce_bucket(void* buf, size_t count, int average, void* stream) {
  CL4_REQUIRE(g_comm, "allreduce_bucket: cl4_comm_init has not been called");
  CL4_REQUIRE(buf && count > 0, "allreduce_bucket: empty bucket");
  return check(g_rccl.AllReduce(buf, buf, count, NCCL_FLOAT32, average ? NCCL_AVG : NCCL_SUM, g_comm, cl4::as_stream(stream)),
               "ncclAllReduce");
}

extern "C" int cl4_allreduce_f64(void* buf, size_t count, void* stream) {
  CL4_REQUIRE(g_comm, "allreduce_f64: cl4_comm_init has not been called");
  CL4_REQUIRE(buf && count > 0, "allreduce_f64: empty vector");
  return check(g_rccl.AllReduce(buf, buf, count, NCCL_FLOAT64, NCCL_SUM, g_comm, cl4::as_stream(stream)), "ncclAllReduce(f64)");
}

extern "C" int cl4_comm_destroy(void) {
  if (!g_comm) return CL4_OK;
  int rc = check(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  return rc;
}
