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
  // event pairs recorded as EXTERNAL event-record nodes of a hipGraph under capture: every replay of that graph re-records them,
  // egt_prof_collect_graph() reads the latest completed replay.  They belong to the captured graph: never pooled, never dropped.
  std::vector<std::pair<hipEvent_t, hipEvent_t>> graph_pairs;
  int64_t count = 0;
  double ms = 0.0;
  int64_t seen = 0;   // launches met while enabled (egt_prof_stride samples them)
  unsigned long long cap_id = 0;   // the stream capture the counter below belongs to
  int64_t cap_seen = 0;            // launches of this kernel met inside that capture: the stride samples a captured graph from its
                                   // own first launch (launch 0, stride, 2 stride, ...), whatever ran eagerly before the capture
};
std::mutex g_mu;
int g_enabled = 0;
std::string g_filter;  // when non-empty only this kernel is timed
int g_stride = 1;      // time every g_stride-th launch of a timed kernel
std::unordered_map<std::string, ProfEntry> g_prof;
std::vector<hipEvent_t> g_pool;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_retired;   // graph-resident pairs of a reset profile: their graphs may still replay

bool stream_capturing(hipStream_t s, unsigned long long* id = nullptr) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2(s, &st, &cid, &graph, &deps, &ndeps) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (id) *id = cid;
  return st == hipStreamCaptureStatusActive;
}
// An event-record NODE at the capturing stream's current position.  (hipEventRecordWithFlags(.., hipEventRecordExternal) is the
// one-call form of this, but the HIP runtime torch 2.10 ships answers it with hipErrorInvalidValue under capture -- probed on the
// box, tools/micro/graph_event_probe.py; the explicit form below works there: every replay re-records the event.)
bool record_node_in_capture(hipStream_t s, hipEvent_t ev) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps) != hipSuccess || st != hipStreamCaptureStatusActive) { (void)hipGetLastError(); return false; }
  hipGraphNode_t node = nullptr;
  if (hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies) != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}

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
  unsigned long long cid = 0;
  const bool cap = stream_capturing(s, &cid);
  if (cap) {
    if (cid != e.cap_id) { e.cap_id = cid; e.cap_seen = 0; }
    if (g_stride > 1 && (e.cap_seen++ % g_stride) != 0) return;
  } else if (g_stride > 1 && (e.seen++ % g_stride) != 0) return;
  if (cap) {   // the launch is being captured into a hipGraph: the events become event-record nodes of that graph
    hipEvent_t a = get_event(), b = get_event();   // (from the pool egt_prof_enable filled before the capture began)
    if (!record_node_in_capture(s, a)) {   // no node, no pair: a pair that is never recorded would fail every later read
      g_pool.push_back(a); g_pool.push_back(b);
      return;
    }
    e.graph_pairs.emplace_back(a, b);
    *tok = (void*)b;
    return;
  }
  hipEvent_t a = get_event(), b = get_event();
  (void)hipEventRecord(a, s);
  e.pending.emplace_back(a, b);
  *tok = (void*)b;
}

void egt_prof_end(void* tok, hipStream_t s) {
  if (!tok) return;
  if (stream_capturing(s)) (void)record_node_in_capture(s, (hipEvent_t)tok);
  else (void)hipEventRecord((hipEvent_t)tok, s);
}

extern "C" int egt_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on ? 1 : 0;
  if (on) {   // events a capture may need exist before it begins (no event creation inside a capturing region)
    while (g_pool.size() < 256) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; g_pool.push_back(e); }
  }
  if (on == 2) {  // reset: counts, sums and un-read eager pairs go; the pairs that live inside captured graphs stay with their kernels
    for (auto it = g_prof.begin(); it != g_prof.end();) {
      auto& e = it->second;
      for (auto& p : e.pending) { g_pool.push_back(p.first); g_pool.push_back(p.second); }
      e.pending.clear();
      e.count = 0; e.ms = 0.0; e.seen = 0;
      if (e.graph_pairs.empty()) it = g_prof.erase(it); else ++it;
    }
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
    else (void)hipGetLastError();   // a failed read must not leave the thread's sticky error for the next launch check
    g_pool.push_back(p.first);
    g_pool.push_back(p.second);
  }
  e.pending.clear();
  *count = e.count;
  *total_ms = e.ms;
  return EGT_OK;
}

// Timings of the launches that were captured into hipGraphs while the profile was enabled (their hipEvents are external
// event-record nodes of those graphs): adds the LATEST replay's elapsed time of every such pair to the kernel's count / sum.
// Call it when the replay to be read has completed (after a stream synchronisation); returns the number of pairs read.
// `reset_counts` != 0 first zeroes the counts / sums of every kernel (the pairs stay: they belong to the graphs).
extern "C" int egt_prof_collect_graph(int reset_counts) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto& kv : g_prof) {
    auto& e = kv.second;
    if (reset_counts) { e.count = 0; e.ms = 0.0; }
    for (auto& p : e.graph_pairs) {
      float ms = 0.f;
      if (hipEventQuery(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess && ms > 0.f) {
        e.ms += ms; e.count += 1; ++n;
      } else {
        (void)hipGetLastError();   // never-recorded pair (collect before the first replay): legitimate, and not the next launch's error
      }
    }
  }
  return n;
}

// Forget the event pairs of captured graphs (call when those graphs are destroyed or no longer of interest; the events themselves
// are kept alive -- a graph that still replays would otherwise record into freed events).
extern "C" int egt_prof_forget_graphs(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_prof) {
    for (auto& p : kv.second.graph_pairs) g_retired.push_back(p);
    kv.second.graph_pairs.clear();
  }
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
