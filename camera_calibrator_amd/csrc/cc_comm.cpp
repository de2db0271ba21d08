// cc_comm.cpp -- RCCL all-reduce of the reduced normal-equation blocks (one process per GPU).
// RCCL is resolved with dlopen at first use so that single-GPU callers never load it; if the host
// process already imported torch, the loader hands back torch's own librccl.so.1 (same soname).
#include <dlfcn.h>

#include <cstring>
#include <string>

#include "cc_common.hpp"

namespace cc {

namespace {
typedef struct { char internal[128]; } NcclUniqueId;
typedef void* NcclComm;
struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

int load_api() {
  if (g_api.lib) return CC_OK;
  // RCCL must match the HIP runtime this process actually runs on: a Python process that imported
  // torch runs on torch's bundled libamdhip64/librccl, a plain C++ process on /opt/rocm's. Look for
  // librccl next to the libamdhip64 that provides hipGetDeviceCount, then fall back to the soname.
  void* lib = nullptr;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash);
      for (const char* name : {"/librccl.so.1", "/librccl.so"}) {
        lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
      }
    }
  }
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(CC_ERR_COMM, "cannot load librccl: %s", dlerror());
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommDestroy || !g_api.AllReduce)
    return fail(CC_ERR_COMM, "librccl is missing nccl entry points");
  g_api.lib = lib;
  return CC_OK;
}
const char* errstr(int rc) { return g_api.GetErrorString ? g_api.GetErrorString(rc) : "?"; }
constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;
}  // namespace

struct Comm {
  NcclComm comm = nullptr;
  int rank = 0, nranks = 1;
};

int comm_create(const uint8_t id[128], int rank, int nranks, Comm** out) {
  if (int rc = load_api()) return rc;
  NcclUniqueId uid;
  std::memcpy(uid.internal, id, 128);
  Comm* c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  const int rc = g_api.CommInitRank(&c->comm, nranks, uid, rank);
  if (rc != 0) {
    delete c;
    return fail(CC_ERR_COMM, "ncclCommInitRank failed: %s", errstr(rc));
  }
  *out = c;
  return CC_OK;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->comm && g_api.CommDestroy) g_api.CommDestroy(c->comm);
  delete c;
}

int comm_allreduce_sum(Comm* c, double* buf, int n, hipStream_t stream) {
  const int rc = g_api.AllReduce(buf, buf, (size_t)n, kNcclFloat64, kNcclSum, c->comm, stream);
  if (rc != 0) return fail(CC_ERR_COMM, "ncclAllReduce failed: %s", errstr(rc));
  return CC_OK;
}

}  // namespace cc

extern "C" int cc_comm_get_unique_id(uint8_t id[128]) {
  using namespace cc;
  if (!id) return fail(CC_ERR_BAD_ARGUMENT, "cc_comm_get_unique_id: NULL");
  if (int rc = load_api()) return rc;
  NcclUniqueId uid;
  const int rc = g_api.GetUniqueId(&uid);
  if (rc != 0) return fail(CC_ERR_COMM, "ncclGetUniqueId failed: %s", errstr(rc));
  std::memcpy(id, uid.internal, 128);
  return CC_OK;
}
