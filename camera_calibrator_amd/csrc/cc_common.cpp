// cc_common.cpp -- error plumbing, option defaults and host-only helpers of the C ABI.
#include "cc_common.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace cc {

std::string& last_error() {
  static thread_local std::string s;
  return s;
}

double* last_timing() {
  static thread_local double t[5] = {0, 0, 0, 0, 0};
  return t;
}

int fail(int code, const char* fmt, ...) {
  char buf[4096];   // (a stalled exchange describes every rank's mailbox slots)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

int select_device(int device) {
  static std::mutex mu;
  static std::vector<char> is_gfx950;   // per device: 0 unknown, 1 yes (the property query is slow)
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(CC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(CC_ERR_BAD_ARGUMENT, "device %d out of range (0..%d)", device, n - 1);
  bool known;
  {
    std::lock_guard<std::mutex> lk(mu);
    if ((int)is_gfx950.size() < n) is_gfx950.resize((size_t)n, 0);
    known = is_gfx950[(size_t)device] != 0;
  }
  if (!known) {
    hipDeviceProp_t prop;
    CC_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      return fail(CC_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    std::lock_guard<std::mutex> lk(mu);
    is_gfx950[(size_t)device] = 1;
  }
  CC_HIP(hipSetDevice(device));
  return CC_OK;
}

namespace {
std::mutex g_cache_mu;
std::vector<void*> g_pinned_free;
std::vector<std::vector<hipStream_t>> g_stream_free;   // per device
}  // namespace

void* pinned_block_get() {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (!g_pinned_free.empty()) { void* p = g_pinned_free.back(); g_pinned_free.pop_back(); return p; }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, 512, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void pinned_block_put(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_pinned_free.push_back(p);
}

int stream_get(int device, hipStream_t* out) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((int)g_stream_free.size() > device && !g_stream_free[(size_t)device].empty()) {
      *out = g_stream_free[(size_t)device].back();
      g_stream_free[(size_t)device].pop_back();
      return CC_OK;
    }
  }
  CC_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return CC_OK;
}

namespace {
struct Scratch { void* p = nullptr; size_t bytes = 0; bool busy = false; };
std::vector<Scratch> g_scratch;   // per device, guarded by g_cache_mu
constexpr size_t kScratchKeep = (size_t)64 << 20;
}  // namespace

// (a piece that has to grow is marked busy and its old block detached UNDER the lock; hipFree / hipMalloc --
// device-wide waits -- run outside it, so one thread growing a piece no longer stalls every other thread's handle creation and
// teardown behind another thread's persistent solve)
int scratch_get(int device, size_t bytes, void** out, bool* cached) {
  *cached = false;
  if (bytes <= kScratchKeep) {
    void* old = nullptr;
    bool grow = false, mine = false;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      if ((int)g_scratch.size() <= device) g_scratch.resize((size_t)device + 1);
      Scratch& s = g_scratch[(size_t)device];
      if (!s.busy) {
        mine = true;
        s.busy = true;
        if (s.bytes < bytes) { grow = true; old = s.p; s.p = nullptr; s.bytes = 0; }
        else { *out = s.p; *cached = true; return CC_OK; }
      }
    }
    if (mine && grow) {
      if (old) (void)hipFree(old);
      void* fresh = nullptr;
      const hipError_t e = hipMalloc(&fresh, bytes);
      std::lock_guard<std::mutex> lk(g_cache_mu);
      Scratch& s = g_scratch[(size_t)device];
      if (e != hipSuccess) { (void)hipGetLastError(); s.busy = false; return fail(CC_ERR_HIP, "hipMalloc(%zu) failed", bytes); }
      s.p = fresh; s.bytes = bytes;
      *out = fresh; *cached = true;
      return CC_OK;
    }
  }
  CC_HIP(hipMalloc(out, bytes));
  return CC_OK;
}

void scratch_put(int device, void* p, size_t, bool cached) {
  if (!p) return;
  if (cached) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((int)g_scratch.size() > device && g_scratch[(size_t)device].p == p) { g_scratch[(size_t)device].busy = false; return; }
  }
  hipFree(p);
}

// ---- device arena of a solver handle and pinned host staging of the one-shot calls: a few cached pieces each (per device /
// per process), grow-only, handed to one owner at a time -- a caller that re-estimates as images arrive (the reference's
// workflow builds a fresh Calibrator per call: cam_calibration.py:290-322) pays hipMalloc / hipHostMalloc once, not per call,
// and so do up to four host threads calling at once (hipFree / hipHostFree wait for the whole device: from a second thread
// they can stall behind another thread's persistent solve).
namespace {
struct Piece { void* p = nullptr; size_t bytes = 0; bool busy = false; };
constexpr int kPiecesKept = 4;
std::vector<std::vector<Piece>> g_arena;   // per device
std::vector<Piece> g_staging;              // pinned host memory
constexpr size_t kArenaKeep = (size_t)1 << 30, kStagingKeep = (size_t)1 << 30;

// an idle piece for `bytes`: the smallest one that is large enough, else the largest idle one (grown by the caller), else
// a new slot while fewer than kPiecesKept exist; nullptr: all busy (the caller allocates a piece of its own)
Piece* pick_piece(std::vector<Piece>& v, size_t bytes) {
  Piece* fit = nullptr;
  Piece* grow = nullptr;
  for (Piece& x : v) {
    if (x.busy) continue;
    if (x.bytes >= bytes) { if (!fit || x.bytes < fit->bytes) fit = &x; }
    else if (!grow || x.bytes > grow->bytes) grow = &x;
  }
  if (fit) return fit;
  if (grow) return grow;
  if ((int)v.size() < kPiecesKept) { v.emplace_back(); return &v.back(); }
  return nullptr;
}
}  // namespace

int arena_get(int device, size_t bytes, void** out, bool* cached) {
  *cached = false;
  if (bytes <= kArenaKeep) {
    void* old = nullptr;
    size_t slot = 0;
    bool grow = false, mine = false;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      if ((int)g_arena.size() <= device) g_arena.resize((size_t)device + 1);
      auto& v = g_arena[(size_t)device];
      if (v.capacity() < (size_t)kPiecesKept) v.reserve(kPiecesKept);
      if (Piece* a = pick_piece(v, bytes)) {
        mine = true;
        a->busy = true;
        slot = (size_t)(a - v.data());
        if (a->bytes < bytes) { grow = true; old = a->p; a->p = nullptr; a->bytes = 0; }
        else { *out = a->p; *cached = true; return CC_OK; }
      }
    }
    if (mine && grow) {
      if (old) (void)hipFree(old);
      const size_t want = bytes + bytes / 4;   // (room for a problem that grows by a few frames per call)
      void* fresh = nullptr;
      const hipError_t e = hipMalloc(&fresh, want);
      std::lock_guard<std::mutex> lk(g_cache_mu);
      Piece& a = g_arena[(size_t)device][slot];
      if (e != hipSuccess) { (void)hipGetLastError(); a.busy = false; return fail(CC_ERR_HIP, "hipMalloc(%zu) failed", want); }
      a.p = fresh; a.bytes = want;
      *out = fresh; *cached = true;
      return CC_OK;
    }
  }
  CC_HIP(hipMalloc(out, bytes));
  return CC_OK;
}

void arena_put(int device, void* p, bool cached) {
  if (!p) return;
  if (cached) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((int)g_arena.size() > device)
      for (Piece& a : g_arena[(size_t)device])
        if (a.p == p) { a.busy = false; return; }
  }
  (void)hipFree(p);
}

void* staging_get(size_t bytes, bool* cached) {
  *cached = false;
  if (bytes <= kStagingKeep) {
    void* old = nullptr;
    size_t slot = 0;
    bool grow = false, mine = false;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      if (g_staging.capacity() < (size_t)kPiecesKept) g_staging.reserve(kPiecesKept);
      if (Piece* st = pick_piece(g_staging, bytes)) {
        mine = true;
        st->busy = true;
        slot = (size_t)(st - g_staging.data());
        if (st->bytes < bytes) { grow = true; old = st->p; st->p = nullptr; st->bytes = 0; }
        else { *cached = true; return st->p; }
      }
    }
    if (mine && grow) {
      if (old) (void)hipHostFree(old);
      const size_t want = bytes + bytes / 4;
      void* fresh = nullptr;
      const hipError_t e = hipHostMalloc(&fresh, want, hipHostMallocDefault);
      std::lock_guard<std::mutex> lk(g_cache_mu);
      Piece& st = g_staging[slot];
      if (e != hipSuccess) { (void)hipGetLastError(); st.busy = false; return nullptr; }
      st.p = fresh; st.bytes = want;
      *cached = true;
      return fresh;
    }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void staging_put(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (Piece& st : g_staging)
      if (st.p == p) { st.busy = false; return; }
  }
  (void)hipHostFree(p);
}

namespace {
struct PoolBlock { void* p; size_t bytes; };
std::vector<std::vector<PoolBlock>> g_pool;   // per device, guarded by g_cache_mu
std::vector<size_t> g_pool_bytes;
constexpr size_t kPoolKeep = (size_t)8 << 30;
}  // namespace

int pool_alloc(int device, size_t bytes, void** out, size_t* got) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((int)g_pool.size() > device) {
      auto& v = g_pool[(size_t)device];
      int best = -1;
      for (int i = 0; i < (int)v.size(); ++i)
        if (v[(size_t)i].bytes >= bytes && v[(size_t)i].bytes <= bytes + bytes / 4 + ((size_t)1 << 20) && (best < 0 || v[(size_t)i].bytes < v[(size_t)best].bytes)) best = i;
      if (best >= 0) {
        *out = v[(size_t)best].p; *got = v[(size_t)best].bytes;
        g_pool_bytes[(size_t)device] -= v[(size_t)best].bytes;
        v.erase(v.begin() + best);
        return CC_OK;
      }
    }
  }
  CC_HIP(hipMalloc(out, bytes));
  *got = bytes;
  return CC_OK;
}

void pool_free(int device, void* p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if ((int)g_pool.size() <= device) { g_pool.resize((size_t)device + 1); g_pool_bytes.resize((size_t)device + 1, 0); }
    if (g_pool_bytes[(size_t)device] + bytes <= kPoolKeep && g_pool[(size_t)device].size() < 256) {
      g_pool[(size_t)device].push_back(PoolBlock{p, bytes});
      g_pool_bytes[(size_t)device] += bytes;
      return;
    }
  }
  (void)hipFree(p);
}

std::mutex& persist_mutex(int device) {
  static std::mutex table[64];
  return table[device >= 0 && device < 64 ? device : 0];
}

namespace {
struct PersistBackoff {
  int level = 0;            // give-ups in a row (0: no window)
  int64_t calls_left = 0;   // solves still to be turned away
  std::chrono::steady_clock::time_point until{};
  bool probing = false;     // one solve is out probing: the others keep to the several-kernel form until it reports
};
std::mutex g_backoff_mu;
PersistBackoff g_backoff[64][2];
PersistBackoff& backoff_of(int device, int kind) { return g_backoff[device >= 0 && device < 64 ? device : 0][kind ? 1 : 0]; }
}  // namespace

bool persist_test_drop_control(const char* env_name, int* remaining) {
  std::lock_guard<std::mutex> lk(g_backoff_mu);
  if (*remaining == -2) {
    const char* e = getenv(env_name);
    if (!e || !*e || std::strcmp(e, "0") == 0) *remaining = 0;
    else if (std::strncmp(e, "first", 5) == 0) *remaining = std::max(0, atoi(e + 5));
    else *remaining = -1;   // every launch
  }
  if (*remaining == -1) return true;
  if (*remaining > 0) { --*remaining; return true; }
  return false;
}

bool persist_device_try(int device, int kind) {
  std::lock_guard<std::mutex> lk(g_backoff_mu);
  PersistBackoff& b = backoff_of(device, kind);
  if (b.level == 0) return true;
  if (b.probing) return false;
  if (b.calls_left > 0) --b.calls_left;
  if (b.calls_left > 0 || std::chrono::steady_clock::now() < b.until) return false;
  b.probing = true;
  return true;
}

void persist_device_gave_up(int device, int kind) {
  static const char* e_calls = getenv("CC_PERSIST_BACKOFF_CALLS");     // (tests: a window short enough to watch the re-probe)
  static const char* e_ms = getenv("CC_PERSIST_BACKOFF_MS");
  std::lock_guard<std::mutex> lk(g_backoff_mu);
  PersistBackoff& b = backoff_of(device, kind);
  b.level = std::min(b.level + 1, 8);
  b.probing = false;
  const int64_t calls0 = e_calls ? std::max(1, atoi(e_calls)) : 8;
  const int64_t ms0 = e_ms ? std::max(0, atoi(e_ms)) : 2000;
  b.calls_left = std::min<int64_t>(calls0 << (b.level - 1), 1024) + 1;   // (+ 1: persist_device_try counts before it looks)
  b.until = std::chrono::steady_clock::now() + std::chrono::milliseconds(std::min<int64_t>(ms0 << (b.level - 1), 600000));
}

void persist_device_completed(int device, int kind) {
  std::lock_guard<std::mutex> lk(g_backoff_mu);
  PersistBackoff& b = backoff_of(device, kind);
  if (b.level != 0) b = PersistBackoff{};
}

// Solver form / reruns of the handles this thread's last one-shot call worked with (the call destroys them before it returns:
// cc_last_call_solver_status is how its caller -- the C++ classes included -- learns that a persistent solve gave up and was run
// again in the several-kernel form: 42 ms to 1.3 s late, results equal to rounding only).
namespace {
thread_local int t_last_form = 0, t_last_reruns = 0;
thread_local std::string t_last_note;
}  // namespace
void last_call_status_reset() { t_last_form = 0; t_last_reruns = 0; t_last_note.clear(); }
void last_call_status_record(int form, int reruns, const std::string& note) {
  if (reruns > 0 || t_last_reruns == 0) { t_last_form = form; if (reruns > 0) t_last_note = note; }
  t_last_reruns += reruns;
}

int parallel_parts(int64_t n, int64_t min_per_part) {
  if (n < 2 * min_per_part) return 1;
  const unsigned hw = std::thread::hardware_concurrency();
  const int64_t cap = std::min<int64_t>(16, hw ? (int64_t)hw : 4);
  return (int)std::max<int64_t>(1, std::min<int64_t>(cap, n / min_per_part));
}

// ---- host worker pool (round 5). Every parallel phase of the library (regrouping of the observations, per-observation cost
// scatter) and of the C++ classes (cc_parallel_for: ExtrinsicsCalibrator::Optimize's fills) runs on ONE set of threads that
// lives as long as the process: created at first use, grown to the largest part count asked for (at most 15 workers + the
// caller), joined by cc_release_caches. Before, each phase created and joined its own std::threads -- up to 16, two or three
// times per Optimize(). One job at a time (a second host thread waits its turn; the phases are memory-bound copies, running
// two at once on the same cores gains nothing). Task 0 runs on the calling thread, which also helps with the rest.
namespace {
// This thread is inside a job of the pool -- as its caller or as one of its workers. A job started from there (a callback of
// cc_parallel_for that calls cc_parallel_for, or a library entry point that uses the pool) runs INLINE on this thread: job_mu is
// not recursive, and a worker that waited for the pool would wait for itself.
thread_local int t_in_pool_job = 0;
struct InPoolJob {
  InPoolJob() { ++t_in_pool_job; }
  ~InPoolJob() { --t_in_pool_job; }
};
struct WorkerPool {
  std::mutex job_mu;                 // one job at a time
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> th;
  const std::function<void(int)>* fn = nullptr;
  int parts = 0, next = 0, pending = 0;
  unsigned long long gen = 0;
  bool stop = false;
  void worker() {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || (gen != seen && next < parts); });
      if (stop) return;
      seen = gen;
      while (next < parts) {
        const int t = next++;
        const std::function<void(int)>* f = fn;
        lk.unlock();
        { InPoolJob mark; (*f)(t); }
        lk.lock();
        if (--pending <= 0) cv_done.notify_all();
      }
    }
  }
};
WorkerPool& pool() { static WorkerPool* p = new WorkerPool; return *p; }   // (never destroyed: see g_flat_* in extrinsics_calibrator.cpp)
}  // namespace

void parallel_tasks(int parts, const std::function<void(int)>& fn) {
  if (parts <= 1) { fn(0); return; }
  if (t_in_pool_job > 0) {   // nested: no second job, the parts one after the other
    for (int t = 0; t < parts; ++t) fn(t);
    return;
  }
  WorkerPool& P = pool();
  std::lock_guard<std::mutex> job(P.job_mu);
  InPoolJob mark;
  // (an exception out of fn on this thread must not leave the workers with a pointer to a dead job: the guard below waits for
  // the parts already started and clears the job before the stack unwinds further)
  struct JobGuard {
    WorkerPool& P;
    ~JobGuard() {
      std::unique_lock<std::mutex> lk(P.mu);
      P.pending -= P.parts - P.next;   // parts nobody has started: nobody will
      P.next = P.parts;
      P.cv_done.wait(lk, [&] { return P.pending <= 0; });
      P.fn = nullptr; P.parts = 0; P.next = 0; P.pending = 0;
    }
  } guard{P};
  {
    std::unique_lock<std::mutex> lk(P.mu);
    P.stop = false;
    while ((int)P.th.size() < parts - 1 && P.th.size() < 15) P.th.emplace_back([&P] { P.worker(); });
    P.fn = &fn; P.parts = parts; P.next = 1; P.pending = parts - 1;
    ++P.gen;
  }
  P.cv_work.notify_all();
  fn(0);
  std::unique_lock<std::mutex> lk(P.mu);
  while (P.next < P.parts) {   // (the caller takes what no worker has started yet)
    const int t = P.next++;
    --P.pending;   // (before the call: a part that throws on this thread is nobody's to wait for)
    lk.unlock();
    fn(t);
    lk.lock();
  }
  P.cv_done.wait(lk, [&] { return P.pending == 0; });
}

void parallel_pool_release() {
  if (t_in_pool_job > 0) return;   // (cc_release_caches from inside a task of the pool: the threads stay, they are in use)
  WorkerPool& P = pool();
  std::lock_guard<std::mutex> job(P.job_mu);
  std::vector<std::thread> th;
  {
    std::lock_guard<std::mutex> lk(P.mu);
    P.stop = true;
    th.swap(P.th);
  }
  P.cv_work.notify_all();
  for (auto& t : th) t.join();
  std::lock_guard<std::mutex> lk(P.mu);
  P.stop = false;
}

int parallel_pool_threads() {
  WorkerPool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  return (int)P.th.size();
}

void parallel_ranges(int64_t n, int64_t min_per_part, const std::function<void(int, int64_t, int64_t)>& fn) {
  const int parts = parallel_parts(n, min_per_part);
  parallel_tasks(parts, [&](int t) { fn(t, n * t / parts, n * (t + 1) / parts); });
}

void stream_put(int device, hipStream_t s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  if ((int)g_stream_free.size() <= device) g_stream_free.resize((size_t)device + 1);
  g_stream_free[(size_t)device].push_back(s);   // idle (the owner synchronised it) and reusable
}

}  // namespace cc

extern "C" {

void cc_options_init(cc_options* o) {
  o->max_iterations = 100;  // calibrator.cpp:319
  o->use_nonmonotonic_steps = 1;  // calibrator.cpp:315
  o->max_consecutive_nonmonotonic_steps = 5;
  o->jacobi_scaling = 1;
  o->max_consecutive_invalid_steps = 5;
  o->check_interval = 4;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->use_graph = 1;
  o->profile_kernels = 0;
}

const char* cc_last_error(void) { return cc::last_error().c_str(); }

// Pinned host memory for a caller that packs its inputs itself (the C++ classes: one memcpy per frame straight into memory
// hipMemcpyAsync reads at full PCIe rate, cached between calls). Not needed for correctness: every entry point takes any
// host pointer.
void* cc_host_staging_acquire(size_t bytes) {
  bool cached = false;
  return cc::staging_get(bytes ? bytes : 1, &cached);
}
void cc_host_staging_release(void* p) { cc::staging_put(p); }

// Phases of the last cc_intrinsics_estimate / cc_intrinsics_optimize of this thread, wall milliseconds:
// [0] handle + device arena, [1] upload (enqueue + wait), [2] Zhang initialisation incl. its read-back, [3] solve,
// [4] read-back + teardown.
void cc_last_call_timing(double out_ms[5]) {
  for (int i = 0; i < 5; ++i) out_ms[i] = cc::last_timing()[i];
}
// Gives back what the library keeps between calls for the next one: pooled device blocks, the idle device arena and scratch
// piece of every device, the idle pinned staging block (cached streams and the 512-byte control blocks stay: they cost
// nothing). Safe at any time -- a piece a live handle is using is not touched; the next call allocates again.
void cc_release_caches(void) {
  std::vector<std::pair<int, void*>> dev;
  std::vector<void*> pinned;
  {
    std::lock_guard<std::mutex> lk(cc::g_cache_mu);
    for (size_t d = 0; d < cc::g_pool.size(); ++d) {
      for (auto& b : cc::g_pool[d]) dev.emplace_back((int)d, b.p);
      cc::g_pool[d].clear();
      cc::g_pool_bytes[d] = 0;
    }
    for (size_t d = 0; d < cc::g_arena.size(); ++d)
      for (auto& a : cc::g_arena[d])
        if (!a.busy && a.p) { dev.emplace_back((int)d, a.p); a = {}; }
    for (size_t d = 0; d < cc::g_scratch.size(); ++d)
      if (!cc::g_scratch[d].busy && cc::g_scratch[d].p) { dev.emplace_back((int)d, cc::g_scratch[d].p); cc::g_scratch[d] = {}; }
    for (auto& st : cc::g_staging)
      if (!st.busy && st.p) { pinned.push_back(st.p); st = {}; }
  }
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (auto& e : dev) { if (hipSetDevice(e.first) == hipSuccess) (void)hipFree(e.second); }
  if (cur >= 0) (void)hipSetDevice(cur);
  for (void* q : pinned) (void)hipHostFree(q);
  (void)hipGetLastError();
  cc::rig_release_host_caches();
  cc::parallel_pool_release();   // (the host worker threads: the next parallel phase starts them again)
}

void cc_parallel_for(int32_t parts, void (*fn)(void* ctx, int32_t part), void* ctx) {
  if (!fn || parts < 1) return;
  cc::parallel_tasks(parts, [fn, ctx](int t) { fn(ctx, (int32_t)t); });
}
int32_t cc_parallel_parts(int64_t n, int64_t min_per_part) { return cc::parallel_parts(n, min_per_part < 1 ? 1 : min_per_part); }
int32_t cc_host_pool_threads(void) { return cc::parallel_pool_threads(); }

int cc_last_call_solver_status(int32_t* form, int32_t* reruns, char* note, int32_t note_capacity) {
  if (form) *form = cc::t_last_form;
  if (reruns) *reruns = cc::t_last_reruns;
  if (note && note_capacity > 0) std::snprintf(note, (size_t)note_capacity, "%s", cc::t_last_note.c_str());
  return CC_OK;
}
const char* cc_version(void) { return "camera_calibrator_amd 0.1 (gfx950, HIP)"; }

int cc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Contiguous frame ranges with (nearly) equal observation counts: frame f goes to the rank whose
// ideal observation interval contains the midpoint of the frame's observation range.
int cc_partition_frames(int64_t F, const int64_t* off, int32_t nranks, int64_t* first) {
  if (F < 0 || !off || nranks < 1 || !first) return cc::fail(CC_ERR_BAD_ARGUMENT, "cc_partition_frames: bad arguments");
  const int64_t N = off[F];
  int64_t f = 0;
  first[0] = 0;
  for (int r = 1; r < nranks; ++r) {
    // boundary r: first frame whose midpoint is at or beyond N * r / nranks
    const double target = (double)N * r / nranks;
    while (f < F && 0.5 * ((double)off[f] + (double)off[f + 1]) < target) ++f;
    // keep every rank non-empty when there are enough frames
    const int64_t min_f = std::min<int64_t>(F, r);
    const int64_t max_f = std::max<int64_t>(min_f, F - (nranks - r));
    first[r] = std::min(std::max(f, min_f), max_f);
    f = first[r];
  }
  first[nranks] = F;
  return CC_OK;
}

}  // extern "C"
