// A C host of the C-ABI (no torch, no Python): the attention-block stack's forward + backward captured into a hipGraph
// with a device-resident random-mask seed, replayed, and compared BITWISE with eager calls that pass the same seeds
// as host arguments.  Built by egt_amd/build.py (hipcc), run by tests/test_capi_graph_gpu.py on the GPU box.
//   exit code 0 + "OK ..." on success.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "egt_amd.h"

#define HIP_OK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); return 2; } } while (0)
#define EGT_CALL(x) do { int rc__ = (x); if (rc__ != EGT_OK) { printf("egt error %d (%s) at %s:%d\n", rc__, egt_last_error_string(), __FILE__, __LINE__); return 3; } } while (0)

static uint64_t g_lcg = 0x1234567887654321ull;
static float rnd() {   // uniform in [-1, 1)
  g_lcg = g_lcg * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((g_lcg >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

template <typename T>
static T* dev_alloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr; return (T*)p; }

static float* dev_random(size_t n, float scale, float offset) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = offset + scale * rnd();
  float* d = dev_alloc<float>(n);
  if (d) (void)hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  // usage: graph_replay [N [De [B [Ly [timed_replays]]]]]   (timed_replays > 0: also time that many replays with hipEvents)
  const int N = argc > 1 ? atoi(argv[1]) : 48, H = 8, d = 8, De = argc > 2 ? atoi(argv[2]) : 64, B = argc > 3 ? atoi(argv[3]) : 4,
            Ly = argc > 4 ? atoi(argv[4]) : 3, timed = argc > 5 ? atoi(argv[5]) : 0, Dh = H * d;
  const uint64_t S0 = 0x0123456789ABCDEFull, STEP = 0xD1B54A32D192ED03ull;
  const int steps = 3;
  if (egt_abi_version() != EGT_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }

  egt_block_desc desc;
  memset(&desc, 0, sizeof(desc));
  desc.B = B; desc.N = N; desc.H = H; desc.d = d; desc.De = De; desc.dtype = EGT_F32;
  desc.flags = EGT_BF_GATE | EGT_BF_TRAINING | EGT_BF_CLIP;
  desc.clip_lo = -5.f; desc.clip_hi = 5.f; desc.random_mask_prob = 0.2f; desc.ln_eps = 1e-3f;
  if (!egt_block_supported(&desc)) { printf("geometry not covered by the fused block\n"); return 1; }

  // parameters (Keras layouts) and their gradient buffers, in the field order of egt_block_params
  const size_t psz[14] = {(size_t)De, (size_t)De, (size_t)De * H, (size_t)H, (size_t)De * H, (size_t)H, (size_t)Dh, (size_t)Dh,
                          (size_t)Dh * 3 * Dh, (size_t)3 * Dh, (size_t)Dh * Dh, (size_t)Dh, (size_t)H * De, (size_t)De};
  const bool is_gamma[14] = {true, false, false, false, false, false, true, false, false, false, false, false, false, false};
  std::vector<egt_block_params> params(Ly), grads(Ly);
  size_t gtotal = 0;
  for (int i = 0; i < 14; ++i) gtotal += psz[i];
  float* gflat = dev_alloc<float>(gtotal * Ly);
  if (!gflat) return 2;
  for (int l = 0; l < Ly; ++l) {
    const void** pp = reinterpret_cast<const void**>(&params[l]);
    const void** gp = reinterpret_cast<const void**>(&grads[l]);
    size_t off = (size_t)l * gtotal;
    for (int i = 0; i < 14; ++i) {
      pp[i] = dev_random(psz[i], is_gamma[i] ? 0.2f : 0.15f, is_gamma[i] ? 1.0f : 0.0f);
      if (!pp[i]) return 2;
      gp[i] = gflat + off; off += psz[i];
    }
  }
  const size_t hn = (size_t)B * N * Dh, en = (size_t)B * N * N * De;
  float *h = dev_random(hn, 1.f, 0.f), *e = dev_random(en, 1.3f, 0.f), *dh_out = dev_random(hn, 1.f, 0.f), *de_out = dev_random(en, 1.f, 0.f);
  float *h_out = dev_alloc<float>(hn), *e_out = dev_alloc<float>(en), *dh = dev_alloc<float>(hn), *de = dev_alloc<float>(en);
  std::vector<uint8_t> km((size_t)B * N, 1);
  for (int m = N - 5; m < N; ++m) km[(size_t)1 * N + m] = 0;     // graph 1 has 5 padded nodes
  uint8_t* key_mask = dev_alloc<uint8_t>(km.size());
  HIP_OK(hipMemcpy(key_mask, km.data(), km.size(), hipMemcpyHostToDevice));
  const size_t sv_b = egt_stack_saved_bytes(&desc, Ly), ws_b = egt_stack_workspace_bytes(&desc, Ly);
  uint8_t *saved = dev_alloc<uint8_t>(sv_b), *ws = dev_alloc<uint8_t>(ws_b);
  uint64_t* words = dev_alloc<uint64_t>(1);
  if (!h || !e || !dh_out || !de_out || !h_out || !e_out || !dh || !de || !saved || !ws || !words) return 2;

  hipStream_t st;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  struct Snap { std::vector<float> h_out, e_out, dh, de, g; };
  auto snap = [&](Snap& s) -> int {
    s.h_out.resize(hn); s.e_out.resize(en); s.dh.resize(hn); s.de.resize(en); s.g.resize(gtotal * Ly);
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(s.h_out.data(), h_out, hn * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(s.e_out.data(), e_out, en * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(s.dh.data(), dh, hn * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(s.de.data(), de, en * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(s.g.data(), gflat, gtotal * Ly * 4, hipMemcpyDeviceToHost));
    return 0;
  };
  auto step = [&](const egt_block_desc& dd) -> int {
    EGT_CALL(egt_stack_fwd(&dd, Ly, params.data(), h, e, key_mask, nullptr, h_out, e_out, saved, ws, st));
    EGT_CALL(egt_stack_bwd(&dd, Ly, params.data(), h, e, key_mask, nullptr, saved, dh_out, de_out, dh, de, grads.data(), ws, st));
    return 0;
  };

  // eager reference: the seed of step k is a HOST argument
  std::vector<Snap> want(steps);
  for (int k = 0; k < steps; ++k) {
    egt_block_desc dd = desc;
    dd.seed = S0 + (uint64_t)(k + 1) * STEP;
    if (int rc = step(dd)) return rc;
    if (int rc = snap(want[k])) return rc;
  }
  if (memcmp(want[0].h_out.data(), want[1].h_out.data(), hn * 4) == 0) { printf("the random mask did not change between steps\n"); return 1; }

  // captured: seed word in HBM, advanced by a captured launch in front of the forward
  HIP_OK(hipMemcpy(words, &S0, 8, hipMemcpyHostToDevice));
  egt_block_desc dg = desc;
  dg.flags |= EGT_BF_SEED_DEVICE; dg.seed = 0; dg.seed_device = words;
  hipGraph_t graph; hipGraphExec_t exec;
  HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  EGT_CALL(egt_seed_advance(words, 1, STEP, st));
  if (int rc = step(dg)) return rc;
  HIP_OK(hipStreamEndCapture(st, &graph));
  HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  size_t nodes = 0;
  HIP_OK(hipGraphGetNodes(graph, nullptr, &nodes));
  for (int k = 0; k < steps; ++k) {
    HIP_OK(hipMemsetAsync(h_out, 0xFF, hn * 4, st));       // a replay must rewrite everything it returns
    HIP_OK(hipMemsetAsync(de, 0xFF, en * 4, st));
    HIP_OK(hipGraphLaunch(exec, st));
    Snap got;
    if (int rc = snap(got)) return rc;
    const bool same = !memcmp(got.h_out.data(), want[k].h_out.data(), hn * 4) && !memcmp(got.e_out.data(), want[k].e_out.data(), en * 4) &&
                      !memcmp(got.dh.data(), want[k].dh.data(), hn * 4) && !memcmp(got.de.data(), want[k].de.data(), en * 4) &&
                      !memcmp(got.g.data(), want[k].g.data(), gtotal * Ly * 4);
    if (!same) { printf("replay %d differs from the eager call with the same seed\n", k + 1); return 1; }
  }
  uint64_t w = 0;
  HIP_OK(hipMemcpy(&w, words, 8, hipMemcpyDeviceToHost));
  if (w != S0 + (uint64_t)steps * STEP) { printf("seed word %llx\n", (unsigned long long)w); return 1; }
  if (timed > 0) {   // library-only throughput of the step: no Python, no torch, one hipGraphLaunch per step
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (int k = 0; k < 5; ++k) HIP_OK(hipGraphLaunch(exec, st));
    HIP_OK(hipEventRecord(e0, st));
    for (int k = 0; k < timed; ++k) HIP_OK(hipGraphLaunch(exec, st));
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("TIMED %d replays: %.4f ms per step, %.1f graphs/s (C host, hipEvents around the replays)\n", timed, ms / timed, B * 1000.0 * timed / ms);
  }
  printf("OK B=%d N=%d De=%d Ly=%d: %d hipGraph replays (%zu graph nodes) bit-identical to the eager host-seed calls\n", B, N, De, Ly, steps, nodes);
  return 0;
}
