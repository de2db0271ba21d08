// cc_comm.cpp -- RCCL all-reduce of the reduced normal-equation blocks (one process per GPU).
// RCCL is resolved with dlopen at first use so that single-GPU callers never load it; if the host
// process already imported torch, the loader hands back torch's own librccl.so.1 (same soname).
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

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

// ---------------------------------------------------------------------------------------------
// Mailbox exchange, host side: one uncached device allocation per rank, exported through hipIpc and
// mapped by every peer of the node (layout and protocol: cc_device.hpp).
// ---------------------------------------------------------------------------------------------
void mailbox_release(Mailbox* m) {
  for (int r = 0; r < kP2pMaxRanks; ++r) {
    if (m->peer[r] && m->peer_ipc[r]) hipIpcCloseMemHandle(m->peer[r]);
    m->peer[r] = nullptr;
    m->peer_ipc[r] = false;
  }
  if (m->local) { hipFree(m->local); m->local = nullptr; }
  if (m->seq) { hipFree(m->seq); m->seq = nullptr; }
  m->sw[0] = m->sw[1] = 0;
}

int mailbox_alloc(Mailbox* m, int doubles_kind0, int doubles_kind1) {
  mailbox_release(m);
  m->sw[0] = (2 * doubles_kind0 + 63) / 64 * 64;
  m->sw[1] = (2 * doubles_kind1 + 63) / 64 * 64;
  const size_t words = (size_t)2 * kP2pMaxRanks * ((size_t)m->sw[0] + m->sw[1]);
  CC_HIP(hipExtMallocWithFlags((void**)&m->local, words * sizeof(unsigned long long), hipDeviceMallocUncached));
  CC_HIP(hipMemset(m->local, 0, words * sizeof(unsigned long long)));
  CC_HIP(hipMalloc(&m->seq, 2 * sizeof(unsigned long long)));
  CC_HIP(hipMemset(m->seq, 0, 2 * sizeof(unsigned long long)));
  CC_HIP(hipDeviceSynchronize());
  return CC_OK;
}

int mailbox_export(Mailbox* m, int doubles_kind0, int doubles_kind1, uint8_t handle[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size is part of the C ABI");
  if (int rc = mailbox_alloc(m, doubles_kind0, doubles_kind1)) return rc;
  hipIpcMemHandle_t hnd;
  CC_HIP(hipIpcGetMemHandle(&hnd, m->local));
  std::memcpy(handle, &hnd, 64);
  return CC_OK;
}

int mailbox_wire_local(Mailbox* m, int rank, int nranks, Mailbox* const* all, const int* devices, P2pDev* out) {
  if (!m->local) return fail(CC_ERR_STATE, "exchange wiring: allocate the mailbox first");
  CC_HIP(hipSetDevice(devices[rank]));
  for (int r = 0; r < nranks; ++r) {
    if (!all[r]->local || all[r]->sw[0] != m->sw[0] || all[r]->sw[1] != m->sw[1])
      return fail(CC_ERR_STATE, "exchange wiring: rank %d has no mailbox of the same shape", r);
    if (devices[r] != devices[rank]) {
      int can = 0;
      CC_HIP(hipDeviceCanAccessPeer(&can, devices[rank], devices[r]));
      if (!can) return fail(CC_ERR_COMM, "device %d cannot access device %d directly", devices[rank], devices[r]);
      const hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(CC_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", devices[rank], devices[r], hipGetErrorString(e));
      (void)hipGetLastError();
    }
    m->peer[r] = all[r]->local;
    m->peer_ipc[r] = false;
  }
  *out = P2pDev{};
  for (int r = 0; r < kP2pMaxRanks; ++r) out->box[r] = r < nranks ? m->peer[r] : nullptr;
  out->seq = m->seq;
  out->sw[0] = m->sw[0];
  out->sw[1] = m->sw[1];
  out->on = 1;
  return CC_OK;
}

int mailbox_attach(Mailbox* m, int rank, int nranks, const uint8_t* handles, P2pDev* out) {
  if (!m->local) return fail(CC_ERR_STATE, "exchange attach: export the mailbox first");
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { m->peer[r] = m->local; continue; }
    hipIpcMemHandle_t hnd;
    std::memcpy(&hnd, handles + (size_t)r * 64, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, hnd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (int q = 0; q < r; ++q) {
        if (m->peer_ipc[q] && m->peer[q]) hipIpcCloseMemHandle(m->peer[q]);
        m->peer[q] = nullptr;
        m->peer_ipc[q] = false;
      }
      return fail(CC_ERR_COMM, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
    }
    m->peer[r] = static_cast<unsigned long long*>(p);
    m->peer_ipc[r] = true;
  }
  *out = P2pDev{};
  for (int r = 0; r < kP2pMaxRanks; ++r) out->box[r] = r < nranks ? m->peer[r] : nullptr;
  out->seq = m->seq;
  out->sw[0] = m->sw[0];
  out->sw[1] = m->sw[1];
  out->on = 1;
  return CC_OK;
}

// After a wait on the mailboxes gave up: where every rank's posts stand in THIS rank's mailbox, per kind -- the link of
// the chain that stalled is the rank whose words for the epoch last waited for are not there. The slot of an epoch e holds
// e (posted), e - 2 (the previous use of that parity: not posted yet) or a mix (a post in flight). Synchronous copies
// out of the (uncached) mailbox; the caller has drained its stream.
std::string mailbox_describe(const Mailbox* m, int rank, int nranks) {
  if (!m->local || !m->seq) return "no mailbox";
  unsigned long long seq[2] = {0, 0};
  if (hipMemcpy(seq, m->seq, sizeof(seq), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return "mailbox unreadable"; }
  char buf[160];
  std::string out;
  std::snprintf(buf, sizeof(buf), "mailbox of rank %d of %d:", rank, nranks);
  out += buf;
  for (int kind = 0; kind < 2; ++kind) {
    const unsigned long long e = seq[kind];
    std::snprintf(buf, sizeof(buf), " %s, last epoch waited for %llu -", kind == 0 ? "reduced sums (kind 0)" : "; statistics (kind 1)", e);
    out += buf;
    const size_t base = kind == 0 ? 0 : (size_t)2 * kP2pMaxRanks * (size_t)m->sw[0];
    for (int r = 0; r < nranks; ++r) {
      const size_t off = base + ((size_t)(e & 1ull) * kP2pMaxRanks + (size_t)r) * (size_t)m->sw[kind];
      std::vector<unsigned long long> w((size_t)m->sw[kind]);
      if (hipMemcpy(w.data(), m->local + off, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return out + " unreadable"; }
      size_t have = 0, older = 0;
      for (unsigned long long x : w) { have += (x >> 32) == (e & 0xffffffffull) ? 1 : 0; older += (x >> 32) != (e & 0xffffffffull) && x != 0ull ? 1 : 0; }
      if (have && !older) std::snprintf(buf, sizeof(buf), " rank %d posted (%zu words)", r, have);
      else if (have) std::snprintf(buf, sizeof(buf), " rank %d PARTLY posted (%zu words of this epoch, %zu of an older one)", r, have, older);
      else std::snprintf(buf, sizeof(buf), " rank %d MISSING (first word carries epoch %llu)", r, w.empty() ? 0ull : (w[0] >> 32));
      out += buf;
      out += r + 1 < nranks ? "," : "";
    }
  }
  return out;
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
