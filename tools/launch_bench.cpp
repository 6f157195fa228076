// launch_bench.cpp -- per-kernel floor of a dependent chain of trivial kernels (stream vs hipGraph)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void trivial(float* p, int n) { if (n < 0) p[threadIdx.x] = 1.f; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
struct Big { char pad[224]; float* p; int n; };
__global__ void trivial_bigarg(Big b) { if (b.n < 0) b.p[threadIdx.x] = 1.f; }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* d; CK(hipMalloc(&d, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  struct Cfg { const char* name; int blocks, threads, kind; } cfgs[] = {{"trivial 1x64", 1, 64, 0}, {"trivial 256x512", 256, 512, 0}, {"trivial 1024x256", 1024, 256, 0},
      {"touch 256x512 (512 KB rw)", 256, 512, 1}, {"trivial 256x512 232B kernarg", 256, 512, 2}};
  for (auto& c : cfgs) {
    auto launch = [&]() {
      if (c.kind == 0) hipLaunchKernelGGL(trivial, dim3(c.blocks), dim3(c.threads), 0, st, d, 0);
      else if (c.kind == 1) hipLaunchKernelGGL(touch, dim3(c.blocks), dim3(c.threads), 0, st, d, c.blocks * c.threads);
      else { Big b{}; b.p = d; b.n = 0; hipLaunchKernelGGL(trivial_bigarg, dim3(c.blocks), dim3(c.threads), 0, st, b); }
    };
    for (int i = 0; i < 100; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-32s stream: %.2f us/kernel", c.name, ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("   graph(200 nodes): %.2f us/kernel\n", ms * 1e3 / 2000);
  }
  return 0;
}
