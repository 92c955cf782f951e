// Host-side runtime of the pacingpseudo HIP library: version, thread-local error string, and the optional
// per-kernel-family profiler (HIP events recorded on the launch stream around each C-ABI call).
#include "pp_common.h"
#include <stdlib.h>
#include <vector>
#include <mutex>

#define PP_VERSION 600      // round 6 (the ABI changed incompatibly in round 5 without a bump: ADVICE r05); _lib.py checks a minimum

static thread_local char g_err[512] = "";

void pp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pp_last_error(void) { return g_err; }
extern "C" int pp_version(void) { return PP_VERSION; }

extern "C" int pp_device_info(int* cu_count, int* lds_per_cu_kb, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { pp_set_error("hipGetDevice: %s", hipGetErrorString(e)); return (int)e; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) { pp_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return (int)e; }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_per_cu_kb) *lds_per_cu_kb = (int)(prop.maxSharedMemoryPerMultiProcessor / 1024);
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return 0;
}

// ---- numeric mode of the matrix kernels (pp_common.h: pp_f16_products) ----
#include <atomic>
static std::atomic<int> g_products{0};           // 0 = not initialised yet
int pp_f16_products() {
  int n = g_products.load(std::memory_order_relaxed);
  if (n == 0) {
    const char* e = getenv("PP_F16_PRODUCTS");
    n = (e && atoi(e) == 1) ? 1 : 3;
    g_products.store(n, std::memory_order_relaxed);
  }
  return n;
}
extern "C" int pp_set_matrix_products(int n) {
  if (n != 1 && n != 3) { pp_set_error("pp_set_matrix_products: 1 (fp16 operands) or 3 (split operands, fp32 grade)"); return PP_ERR_ARG; }
  g_products.store(n, std::memory_order_relaxed);
  return 0;
}
extern "C" int pp_get_matrix_products(void) { return pp_f16_products(); }

// ---- CU budget of the direct weight-gradient kernels (pp_common.h: pp_wgrad_cus) ----
// Per THREAD (a process-global value let two engines, or a backward on the autograd thread and a direct ABI caller on
// the main thread, see each other's budget): a thread that never set one launches with the default.  Workspace queries
// (pp_conv3x3_bwd_weight_workspace) do not depend on it: they size for the largest budget.
static thread_local int g_wgrad_cus = 0;         // 0 = not set by this thread
int pp_wgrad_cus() {
  if (g_wgrad_cus == 0) {
    static const int dflt = [] { const char* e = getenv("PP_WGRAD_CUS"); return (e && atoi(e) >= 8) ? atoi(e) : 256; }();
    g_wgrad_cus = dflt;
  }
  return g_wgrad_cus;
}
extern "C" int pp_set_wgrad_cus(int cus) {
  if (cus < 8 || cus > PP_WGRAD_CUS_MAX) { pp_set_error("pp_set_wgrad_cus: 8 <= cus <= %d", PP_WGRAD_CUS_MAX); return PP_ERR_ARG; }
  g_wgrad_cus = cus;
  return 0;
}
extern "C" int pp_get_wgrad_cus(void) { return pp_wgrad_cus(); }

// ---- per-(kernel, device) launch attributes ----
#include <map>
static std::mutex g_attr_mu;
static std::map<std::pair<const void*, int>, int> g_attr_done;     // (kernel, device) -> bytes already granted

void pp_max_lds(const void* kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_attr_mu);
  auto key = std::make_pair(kernel, dev);
  auto it = g_attr_done.find(key);
  if (it != g_attr_done.end() && it->second >= bytes) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  g_attr_done[key] = bytes;
}

// ---- profiler ----
struct ProfRec { int kind; double flops, bytes, alg; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static unsigned long long g_prof_mask = ~0ull;
static std::vector<ProfRec> g_recs;           // recorded launches since the last collect
static std::vector<hipEvent_t> g_pool;        // recycled events
static thread_local int g_open = -1;

static hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void pp_prof_begin(int kind, double flops, double bytes, hipStream_t s) { pp_prof_begin2(kind, flops, flops, bytes, s); }

void pp_prof_begin2(int kind, double flops, double alg_flops, double bytes, hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!((g_prof_mask >> kind) & 1ull)) { g_open = -1; return; }
  ProfRec r{kind, flops, bytes, alg_flops, prof_event(), prof_event()};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  g_open = (int)g_recs.size() - 1;
}

void pp_prof_end(hipStream_t s) {
  if (!g_prof_on || g_open < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_open < (int)g_recs.size()) (void)hipEventRecord(g_recs[g_open].b, s);
  g_open = -1;
}

// Restrict the event timing to the kernel families whose bit is set (bit k = PP_KIND_* k).  Two event records per
// launch cost ~3.5 us of queue time each; timing all ~370 launches of a step stretches it by 3.5 % (r02 measurement),
// timing only the ~75 matrix-core launches by well under 1 %.
extern "C" int pp_prof_select(unsigned long long kind_mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_mask = kind_mask;
  return 0;
}

// Pre-create `events` HIP events for the profiler's pool, so that a timed region that follows records into existing events
// instead of calling hipEventCreate per launch (bench.py: 2 events per timed launch).
extern "C" int pp_prof_reserve(int events) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  while ((int)g_pool.size() < events) {
    hipEvent_t e;
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) { pp_set_error("pp_prof_reserve: hipEventCreate: %s", hipGetErrorString(err)); return (int)err; }
    g_pool.push_back(e);
  }
  return 0;
}

extern "C" int pp_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return 0;
}

// out[kind][5] = { launches, total ms, executed flops, algorithmic bytes, algorithmic flops }; clears the record list.
extern "C" int pp_prof_collect(double* out, int kinds) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < kinds * 5; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.kind < kinds) {
      out[r.kind * 5 + 0] += 1.0;
      out[r.kind * 5 + 1] += ms;
      out[r.kind * 5 + 2] += r.flops;
      out[r.kind * 5 + 3] += r.bytes;
      out[r.kind * 5 + 4] += r.alg;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}

// ---- named ranges for rocprofv3 --marker-trace: roctxRangePush / roctxRangePop, resolved at run time
// from librocprofiler-sdk-roctx.so (or the older libroctx64.so) so that the library has no link-time dependency on a
// profiler; without either library the two calls do nothing.
#include <dlfcn.h>
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
static roctx_push_t g_push = nullptr;
static roctx_pop_t g_pop = nullptr;
static std::once_flag g_roctx_once;
static void roctx_resolve() {
  for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
    if (!h) continue;
    g_push = (roctx_push_t)dlsym(h, "roctxRangePushA");
    g_pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
    if (g_push && g_pop) return;
    g_push = nullptr; g_pop = nullptr;
  }
}
extern "C" int pp_range_push(const char* name) {
  std::call_once(g_roctx_once, roctx_resolve);
  return (g_push && name) ? g_push(name) : -1;
}
extern "C" int pp_range_pop(void) {
  std::call_once(g_roctx_once, roctx_resolve);
  return g_pop ? g_pop() : -1;
}
