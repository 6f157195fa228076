// attn_bench.cpp -- stand-alone timing + wave timeline of the fused decode attention (C-ABI, no torch)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dashinfer_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  int B = argc > 1 ? atoi(argv[1]) : 1, L = argc > 2 ? atoi(argv[2]) : 2048, mode = argc > 3 ? atoi(argv[3]) : 0;
  const int n = 28, g = 4, H = 128, S = 128, layers = 28;
  const int max_len = L + 96, spr = (max_len + S - 1) / S;
  hipStream_t st; CK(hipStreamCreate(&st));
  size_t sb = (dihip_span_bytes(g, S, H, mode, DIHIP_BF16) + 255) / 256 * 256;
  size_t nspans = (size_t)layers * B * spr * 2;
  unsigned char* pool; CK(hipMalloc(&pool, nspans * sb));
  { std::vector<uint16_t> h(nspans * sb / 2); for (auto& v : h) v = 0x3c00 | (rand() & 0x83ff); CK(hipMemcpy(pool, h.data(), nspans * sb, hipMemcpyHostToDevice)); }
  std::vector<void*> hk((size_t)layers * B * spr), hv((size_t)layers * B * spr);
  for (size_t i = 0; i < hk.size(); ++i) { hk[i] = pool + ((i * 2 * 7) % nspans) * sb; hv[i] = pool + ((i * 2 * 7 + 7) % nspans) * sb; }
  void **dk, **dv; CK(hipMalloc(&dk, hk.size() * 8)); CK(hipMalloc(&dv, hv.size() * 8));
  CK(hipMemcpy(dk, hk.data(), hk.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice));
  uint16_t* qkv; CK(hipMalloc(&qkv, (size_t)B * (n + 2 * g) * H * 2)); CK(hipMemset(qkv, 0x3c, (size_t)B * (n + 2 * g) * H * 2));
  uint16_t* out; CK(hipMalloc(&out, (size_t)B * n * H * 2));
  std::vector<uint32_t> hl(B, (uint32_t)L); uint32_t* lens; CK(hipMalloc(&lens, B * 4)); CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  std::vector<float> hf(H / 2); for (int i = 0; i < H / 2; ++i) hf[i] = (float)(1.0 / pow(1e6, 2.0 * i / H));
  float *inv, *tab; CK(hipMalloc(&inv, H * 2)); CK(hipMemcpy(inv, hf.data(), H * 2, hipMemcpyHostToDevice)); CK(hipMalloc(&tab, (size_t)(max_len + 1) * H * 4));
  if (dihip_rope_table(st, tab, inv, max_len + 1, H)) { printf("rope_table: %s\n", dihip_last_error()); return 1; }
  size_t wsb = dihip_span_attn_fused_workspace_bytes(B, n, g, H, max_len); void* ws; CK(hipMalloc(&ws, wsb));
  size_t syb = dihip_span_attn_sync_bytes(B, n); void* sync; CK(hipMalloc(&sync, syb)); CK(hipMemset(sync, 0, syb));
  const bool in_launch = !(getenv("MERGE") && getenv("MERGE")[0] == 'l');   // MERGE=launch: the two-launch form
  auto launch = [&](int layer) {
    int rc = dihip_span_attn_decode_fused_sync(st, out, qkv, dk + (size_t)layer * B * spr, dv + (size_t)layer * B * spr, lens, tab, B, n, g, H, S, spr, max_len, mode, DIHIP_BF16,
                                               0.0883883f, ws, wsb, in_launch ? sync : nullptr, in_launch ? syb : 0);
    if (rc) { printf("status %d: %s\n", rc, dihip_last_error()); exit(1); }
  };
  for (int l = 0; l < layers; ++l) launch(l);
  CK(hipStreamSynchronize(st));
  hipGraph_t gr; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < layers; ++l) launch(l);
  CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("fused attention (%s merge), B=%d L=%d kv_mode=%d: %.2f us per layer (graph of %d layers)\n", in_launch ? "in-launch" : "second-launch", B, L, mode,
         ms * 1e3 / (10 * layers), layers);
  // timeline of one launch
  size_t tb = (size_t)64 << 20; unsigned long long* tr; CK(hipMalloc(&tr, tb)); CK(hipMemset(tr, 0, tb));
  for (int l = 0; l < 3; ++l) launch(l);
  CK(hipStreamSynchronize(st));
  dihip_debug_set_trace(tr, tb); launch(5); dihip_debug_set_trace(nullptr, 0);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> ht(tb / 8 / 64); CK(hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; size_t nw = 0;
  for (size_t w = 0; w < ht.size() / 8; ++w) if (ht[w * 8]) { t0 = std::min(t0, ht[w * 8]); nw = w + 1; }
  const char* names[8] = {"entry", "kv loads issued", "q + new token", "token loop done", "records drained", "ticket known", "merged (last)", "end"};
  printf("  %zu waves; stamps (us after first wave start) min / median / max\n", nw);
  for (int s = 0; s < 8; ++s) {
    std::vector<double> v; for (size_t w = 0; w < nw; ++w) if (ht[w * 8 + s]) v.push_back((double)(ht[w * 8 + s] - t0) * 0.01);
    if (v.empty()) continue; std::sort(v.begin(), v.end());
    printf("    %-16s %7.2f %7.2f %7.2f\n", names[s], v.front(), v[v.size() / 2], v.back());
  }
  return 0;
}
