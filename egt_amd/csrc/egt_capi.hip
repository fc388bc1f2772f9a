// C-ABI plumbing: per-thread error string, version, per-kernel HIP-event timing.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "egt_common.h"

static thread_local char g_err[512] = "";

void egt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* egt_last_error_string(void) { return g_err; }
extern "C" int egt_abi_version(void) { return EGT_ABI_VERSION; }

// ---- kernel timing hooks (bench.py's roofline leg) ----------------------------
// When enabled, every launch site brackets its kernel with hipEvents recorded on
// the launch stream; egt_prof_read() resolves them (after the caller has synced).
namespace {
struct ProfEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  int64_t count = 0;
  double ms = 0.0;
  int64_t seen = 0;   // launches met while enabled (egt_prof_stride samples them)
};
std::mutex g_mu;
int g_enabled = 0;
std::string g_filter;  // when non-empty only this kernel is timed
int g_stride = 1;      // time every g_stride-th launch of a timed kernel
std::unordered_map<std::string, ProfEntry> g_prof;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

// ---- EGT_DEBUG_POISON_LDS (egt_common.h) ----------------------------------------
__global__ void __launch_bounds__(256) k_debug_poison_lds() {
  extern __shared__ unsigned poison_sm[];
  for (int i = threadIdx.x; i < 160 * 256; i += 256) poison_sm[i] = 0x7FC00BADu;   // quiet NaN
  __syncthreads();
  if (poison_sm[(threadIdx.x * 37) % (160 * 256)] != 0x7FC00BADu) __builtin_trap();   // (keeps the stores)
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);   // long enough that every CU takes workgroups of this launch
}
int egt_debug_poison_enabled() {
  static const int on = [] { const char* v = getenv("EGT_DEBUG_POISON_LDS"); return (v && v[0] && v[0] != '0') ? 1 : 0; }();
  return on;
}
void egt_debug_poison_lds(hipStream_t s) {
  static bool once = false;
  if (!once) { once = true; (void)hipFuncSetAttribute((const void*)k_debug_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL(k_debug_poison_lds, dim3(1024), dim3(256), 160 * 1024, s);   // one workgroup fills a CU's whole LDS
}

int egt_prof_is_enabled() { return g_enabled; }

void egt_prof_begin(const char* name, hipStream_t s, void** tok) {
  *tok = nullptr;
  if (!g_enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_filter.empty() && g_filter != name) return;
  auto& e = g_prof[name];
  if (g_stride > 1 && (e.seen++ % g_stride) != 0) return;
  hipEvent_t a = get_event(), b = get_event();
  (void)hipEventRecord(a, s);
  e.pending.emplace_back(a, b);
  *tok = (void*)b;
}

void egt_prof_end(void* tok, hipStream_t s) {
  if (tok) (void)hipEventRecord((hipEvent_t)tok, s);
}

extern "C" int egt_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on ? 1 : 0;
  if (on == 2) {  // reset
    for (auto& kv : g_prof) {
      for (auto& p : kv.second.pending) { g_pool.push_back(p.first); g_pool.push_back(p.second); }
    }
    g_prof.clear();
    g_enabled = 1;
  }
  return EGT_OK;
}

// Restrict timing to one kernel (NULL or "" = all): keeps the event overhead out of a
// throughput measurement while still timing the kernel of interest in the same region.
extern "C" int egt_prof_filter(const char* name) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_filter = name ? name : "";
  return EGT_OK;
}

extern "C" int egt_prof_stride(int every) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_stride = every > 1 ? every : 1;
  return EGT_OK;
}

// name == NULL: *count receives the number of distinct kernels; otherwise the
// launch count and summed milliseconds of kernel `name`.  Caller must have
// synchronised the stream(s).
extern "C" int egt_prof_read(const char* name, int64_t* count, double* total_ms) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!count || !total_ms) EGT_FAIL(EGT_E_NULL, "count/total_ms is NULL");
  if (!name) { *count = (int64_t)g_prof.size(); *total_ms = 0; return EGT_OK; }
  auto it = g_prof.find(name);
  if (it == g_prof.end()) { *count = 0; *total_ms = 0; return EGT_OK; }
  auto& e = it->second;
  for (auto& p : e.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { e.ms += ms; e.count += 1; }
    g_pool.push_back(p.first);
    g_pool.push_back(p.second);
  }
  e.pending.clear();
  *count = e.count;
  *total_ms = e.ms;
  return EGT_OK;
}

// kernel names recorded so far, '\n'-separated, into buf
extern "C" int egt_prof_names(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!buf || !cap) EGT_FAIL(EGT_E_NULL, "buf is NULL");
  std::string s;
  for (auto& kv : g_prof) { s += kv.first; s += '\n'; }
  strncpy(buf, s.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return EGT_OK;
}
