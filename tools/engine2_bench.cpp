// engine2_bench.cpp -- what would the persistent LDS-DMA engine buy on THIS layer?  (build it or bury it with data.)
//
// The batch-1 decode layer is five dependent weight-streaming launches; the review asks for the MI355X guide's engine (one loader
// wave per CU feeding an LDS ring by LDS-DMA, consumer waves, 8-byte {data, tag} granule hand-offs between operators) instead.
// Its one advantage over launches is that the loader keeps streaming the NEXT operator's weights into the ring while the CUs
// exchange the previous operator's output.  This tool measures that mechanism on the MLP half of the Qwen2-7B layer -- the two
// largest launches, 72 MB (gate / up) then 36 MB (down) with the 18944-element activation vector exchanged all-to-all between them:
// 24.6 us of the 44.4 us layer in the product -- with the arithmetic replaced by a tunable amount of ALU work per KiB:
//
//   mode 0  two launches   [A: stream 72 MB, publish 74 values per CU]  [B: gather all 18944 values, stream 36 MB]
//   mode 1  one launch     loader runs A's and B's slots back to back through the ring; consumers: A, publish granules, gather
//                          the granules of all CUs (spinning on the tags), B
//
// Geometry as in the guide (MI355X_MICROARCH.md, persistent kernels): one workgroup per CU, 1 loader + 3 consumer waves, ring of
// 8 x 16 KiB slots, fills of 16 x global_load_lds_dwordx4 (nt), granules written with one sc1 store each and swept with sc1 loads.
// Every spin is bounded (a give-up counter is printed): a mistake cannot hang the box.
//   hipcc --offload-arch=gfx950 -O3 -o tools/engine2_bench tools/engine2_bench.cpp && timeout 120 ./tools/engine2_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NSLOT = 7;             // ring slots (7 x 16 KiB + the 37 KB activation vector fit the 160 KB of LDS)
constexpr int SLOT_BYTES = 16384;    // 16 x 1 KiB chunks
constexpr int SPIN_MAX = 1 << 20;    // bounded polls (x s_sleep): ~ a second
#ifndef VM_LAG
#define VM_LAG 32                    // loads that may stay in flight behind a fill (16 per fill; the counter holds 63)
#endif
#ifndef CONS_SLEEP
#define CONS_SLEEP 4                 // s_sleep argument of the consumers' poll of the landed counter (0: spin)
#endif
#ifndef NLOAD
#define NLOAD 1                      // loader waves per CU (slot s belongs to loader s % NLOAD)
#endif
#ifndef USE_COUNTER
#define USE_COUNTER 1                // 1: one arrival counter polled by one lane, then a single sweep; 0: tag-spinning sweeps only
#endif

struct Ctl {                         // LDS control block
  unsigned landed[4];                // per loader: its slots 0 .. landed-1 (local index s / NLOAD) are in the ring
  unsigned freed[NSLOT];             // how often each ring slot has been released
  unsigned bar;                      // consumer barrier counter
  unsigned giveup;
  unsigned all_arrived;              // set by consumer 0 once every CU has published (the others sleep on it)
};

__device__ __forceinline__ unsigned lds_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct Args {
  const unsigned char* w;            // weight region of this launch (A slots of every CU, then B slots of every CU)
  long a_slots, b_slots;             // per CU
  u32x2* granules;                   // [ncu * GPC] {data, tag}
  int gpc;                           // granules per CU
  unsigned epoch;                    // tag of this launch
  unsigned* out;                     // [ncu * 256] sink
  unsigned* gave_up;
  unsigned* arrivals;                // one device-scope counter: CUs that have published this launch's output (monotonic over launches)
  unsigned arrivals_target;          // value of the counter when all CUs of THIS launch have published
  int phase;                         // 0: fused A + exchange + B; 1: A only (+ publish); 2: B only (gather first)
  int work;                          // dummy FMAs per lane per KiB chunk (stand-in for dequantise + MFMA issue)
  unsigned long long* stamps;        // [ncu][8] wall clock (100 MHz): 0 start, 1 first slot landed, 2 A consumed, 3 published, 4 gathered, 5 B consumed, 6 loader done
};
__device__ __forceinline__ unsigned long long wclk() { return __builtin_readcyclecounter() * 0 + wall_clock64(); }

__global__ __launch_bounds__(512) void engine(const Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  Ctl* ctl = reinterpret_cast<Ctl*>(smem + NSLOT * SLOT_BYTES);
  unsigned* act = reinterpret_cast<unsigned*>(smem + NSLOT * SLOT_BYTES + 256);  // gathered data words
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x, ncu = gridDim.x;
  const int NCONS = (int)(blockDim.x >> 6) - NLOAD;  // consumer waves: 3 (the guide's geometry) or 7 (one per remaining wave slot)
  if (threadIdx.x < (int)(sizeof(Ctl) / 4)) reinterpret_cast<unsigned*>(ctl)[threadIdx.x] = 0u;
  __syncthreads();
  unsigned long long* st = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
  if (st && threadIdx.x == 64) st[0] = wclk();
  const long na = a.phase == 2 ? 0 : a.a_slots, nb = a.phase == 1 ? 0 : a.b_slots;
  const long total = na + nb;
  // this CU's streams: A slots contiguous, B slots contiguous (as a GEMV workgroup's tiles are)
  const unsigned char* wa = a.w + (size_t)cu * a.a_slots * SLOT_BYTES;
  const unsigned char* wb = a.w + (size_t)ncu * a.a_slots * SLOT_BYTES + (size_t)cu * a.b_slots * SLOT_BYTES;

  if (wave < NLOAD) {
    // ------------------------------------------------------------------------------------------------ loader(s)
    long mine = 0;  // local index of this loader's next slot
    for (long s = wave; s < total; s += NLOAD, ++mine) {
      const int slot = (int)(s % NSLOT);
      if (s >= NSLOT) {  // wait until the ring slot has been released (s / NSLOT) times
        int spins = 0;
        while (lds_load(&ctl->freed[slot]) < (unsigned)(s / NSLOT)) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > SPIN_MAX) {
            ctl->giveup = 1;
            break;
          }
        }
      }
      const unsigned char* src = (s < na ? wa + (size_t)s * SLOT_BYTES : wb + (size_t)(s - na) * SLOT_BYTES) + lane * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 1024),
                                         (__attribute__((address_space(3))) void*)(ring + slot * SLOT_BYTES + c * 1024), 16, 0, 2 /* nt */);
      // two fills stay in flight; everything older has landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_LAG) : "memory");
      constexpr int BEHIND = VM_LAG / 16;  // fills that may still be in flight
      if (mine >= BEHIND && lane == 0) __hip_atomic_store(&ctl->landed[wave], (unsigned)(mine - BEHIND + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&ctl->landed[wave], (unsigned)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (st && lane == 0) st[6] = wclk();
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int cw = wave - NLOAD;
  u32x4 acc = {0u, 0u, 0u, 0u};
  float f0 = (float)lane, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;
  auto consume = [&](long s0, long s1) {
    for (long s = s0 + cw; s < s1; s += NCONS) {
      int spins = 0;
      while (lds_load(&ctl->landed[s % NLOAD]) <= (unsigned)(s / NLOAD)) {
        if (CONS_SLEEP) __builtin_amdgcn_s_sleep(CONS_SLEEP);
        if (++spins > SPIN_MAX * 8) {
          ctl->giveup = 1;
          break;
        }
      }
      if (st && s == 0 && lane == 0) st[1] = wclk();
      const int slot = (int)(s % NSLOT);
      const u32x4* p = reinterpret_cast<const u32x4*>(ring + slot * SLOT_BYTES) + lane;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        acc ^= p[c * 64];
        for (int k = 0; k < a.work; k += 4) {  // four independent chains: issue-bound like the dequantise / MFMA-feed code, not latency-bound
          f0 = __builtin_fmaf(f0, 1.000001f, 0.5f);
          f1 = __builtin_fmaf(f1, 1.000001f, 0.5f);
          f2 = __builtin_fmaf(f2, 1.000001f, 0.5f);
          f3 = __builtin_fmaf(f3, 1.000001f, 0.5f);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS reads are done before the slot is handed back
      if (lane == 0) __hip_atomic_fetch_add(&ctl->freed[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto cons_barrier = [&](unsigned target) {  // the three consumer waves (the loader does not take part)
    if (lane == 0) __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (lds_load(&ctl->bar) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_MAX) {
        ctl->giveup = 1;
        break;
      }
    }
  };
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.granules, 0, ncu * a.gpc * 8, 0x00020000);
  auto publish = [&]() {  // this CU's output: gpc granules, one sc1 (write-through) store each, then ONE arrival (consumer 0)
    if (cw == 0) {
      if (lane < a.gpc) {
        const u32x2 g = {acc[0] ^ (unsigned)lane, a.epoch};
        __builtin_amdgcn_raw_buffer_store_b64(g, rsrc, (cu * a.gpc + lane) * 8, 0, 16 /* sc1 */);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(a.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto wait_all = [&]() {  // ONE lane of the CU polls the arrival counter (the guide: poll from one lane, with s_sleep); the others sleep on LDS
    if (cw == 0) {
      if (lane == 0) {
        int spins = 0;
        while ((int)(__hip_atomic_load(a.arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.arrivals_target) < 0) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > SPIN_MAX) {
            ctl->giveup = 1;
            break;
          }
        }
        __hip_atomic_store(&ctl->all_arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    int spins = 0;
    while (lds_load(&ctl->all_arrived) == 0u) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > SPIN_MAX) {
        ctl->giveup = 1;
        break;
      }
    }
  };
  auto gather = [&]() {  // all granules of all CUs, swept by the three consumers; a granule is valid when its tag is this epoch
    const int n = ncu * a.gpc;
    for (int base = cw * 64 * 16; base < n; base += NCONS * 64 * 16) {  // a pass: 16 loads per lane in flight
      unsigned ok;
      int spins = 0;
      do {
        u32x2 g[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = base + j * 64 + lane;
          g[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (i < n ? i : n - 1) * 8, 0, 16 /* sc1 */);
        }
        ok = 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = base + j * 64 + lane;
          if (i < n) {
            ok &= g[j][1] == a.epoch;
            act[i] = g[j][0];
          }
        }
        ok = __builtin_amdgcn_readfirstlane(__ballot(ok) == ~0ull);
        if (!ok) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > SPIN_MAX / 16) {
            ctl->giveup = 1;
            break;
          }
        }
      } while (!ok);
    }
  };

  if (a.phase == 2) {  // B as its own launch: the activation vector comes from memory first (what the GEMV prologue does today)
    gather();
    cons_barrier(NCONS);
    consume(0, nb);
  } else {
    consume(0, na);
    cons_barrier(NCONS);  // the CU's A output is complete
    if (st && cw == 0 && lane == 0) st[2] = wclk();
    publish();
    if (st && cw == 0 && lane == 0) st[3] = wclk();
    if (a.phase == 0) {
      if (USE_COUNTER) wait_all();
      gather();
      cons_barrier(2 * NCONS);
      if (st && cw == 0 && lane == 0) st[4] = wclk();
      for (int i = lane; i < 64; i += 64) acc[1] ^= act[(i * 37) % (ncu * a.gpc)];
      consume(na, total);
      cons_barrier(3 * NCONS);
      if (st && cw == 0 && lane == 0) st[5] = wclk();
    }
  }
  a.out[(size_t)cu * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ __float_as_uint(f0 + f1 + f2 + f3);
  if (threadIdx.x == 64 && ctl->giveup) atomicAdd(a.gave_up, 1u);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int layers = 24;
  // Qwen2-7B int4 g128: gate + up 72.2 MB, down 36.1 MB per layer -> per CU slots of 16 KiB
  const long a_slots = (long)((72.2e6 / ncu + SLOT_BYTES - 1) / SLOT_BYTES), b_slots = (long)((36.1e6 / ncu + SLOT_BYTES - 1) / SLOT_BYTES);
  const int gpc = (18944 / 2 + ncu - 1) / ncu;  // granules (two bf16 each) per CU
  const size_t per_layer = (size_t)ncu * (a_slots + b_slots) * SLOT_BYTES;
  unsigned char* w;
  CK(hipMalloc(&w, per_layer * layers));
  CK(hipMemset(w, 0x3c, per_layer * layers));
  u32x2* gran;
  CK(hipMalloc(&gran, (size_t)ncu * gpc * 8));
  CK(hipMemset(gran, 0, (size_t)ncu * gpc * 8));
  unsigned *out, *gave;
  CK(hipMalloc(&out, (size_t)ncu * 512 * 4));
  CK(hipMalloc(&gave, 4));
  CK(hipMemset(gave, 0, 4));
  unsigned* arrivals;
  CK(hipMalloc(&arrivals, 4));
  CK(hipMemset(arrivals, 0, 4));
  unsigned arrived = 0;  // host mirror of the counter: every A (phase 0 or 1) adds ncu
  const size_t lds = NSLOT * SLOT_BYTES + 256 + (size_t)ncu * gpc * 4 + 64;
  CK(hipFuncSetAttribute((const void*)engine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  printf("NLOAD %d VM_LAG %d CONS_SLEEP %d USE_COUNTER %d | ", NLOAD, VM_LAG, CONS_SLEEP, USE_COUNTER);
  printf("%d CUs; per CU %ld + %ld slots of 16 KiB (%.1f + %.1f MB per layer), %d granules per CU (%.1f KB gathered per CU), LDS %zu B\n", ncu, a_slots,
         b_slots, ncu * a_slots * SLOT_BYTES / 1e6, ncu * b_slots * SLOT_BYTES / 1e6, gpc, ncu * gpc * 8 / 1e3, lds);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)ncu * 8 * 8));
  unsigned epoch = 1;
  for (int threads : {256, 512})
  for (int work : {0, 40, 76}) {
    double us[2];
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int l = 0; l < layers; ++l) {
          arrived += (unsigned)ncu;
          Args a{w + per_layer * l, a_slots, b_slots, gran, gpc, epoch++, out, gave, arrivals, arrived, 0, work, nullptr};
          if (mode == 1) {
            a.phase = 0;
            hipLaunchKernelGGL(engine, dim3(ncu), dim3(threads), lds, 0, a);
          } else {
            a.phase = 1;
            hipLaunchKernelGGL(engine, dim3(ncu), dim3(threads), lds, 0, a);
            a.phase = 2;
            hipLaunchKernelGGL(engine, dim3(ncu), dim3(threads), lds, 0, a);
          }
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      us[mode] = best * 1e3 / layers;
    }
    unsigned g = 0;
    CK(hipMemcpy(&g, gave, 4, hipMemcpyDeviceToHost));
    printf("%d loader(s) + %d consumers, work %3d FMA per lane per KiB: two launches %6.2f us | one persistent launch %6.2f us | ratio %.3f | pure stream at 6.5 TB/s %.2f us | gave up %u\n",
           NLOAD, threads / 64 - NLOAD, work, us[0], us[1], us[1] / us[0], per_layer / 6.5e12 * 1e6, g);
  }
  // timeline of ONE fused launch: per CU stamps relative to the earliest start
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int tl_threads = cfg == 0 ? 256 : 512, tl_work = cfg == 2 ? 40 : 0;
    printf("timeline of one persistent launch, %d loader(s) + %d consumers, work %d:\n", NLOAD, tl_threads / 64 - NLOAD, tl_work);
    CK(hipMemset(stamps, 0, (size_t)ncu * 64));
    arrived += (unsigned)ncu;
    Args a{w + per_layer * (cfg + 1), a_slots, b_slots, gran, gpc, epoch++, out, gave, arrivals, arrived, 0, tl_work, stamps};
    hipLaunchKernelGGL(engine, dim3(ncu), dim3(tl_threads), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)ncu * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int c = 0; c < ncu; ++c) t0 = h[c * 8] < t0 ? h[c * 8] : t0;
    const char* names[7] = {"start", "first slot landed", "A consumed", "published", "gathered", "B consumed", "loader done"};
    for (int k = 0; k < 7; ++k) {
      std::vector<double> v;
      for (int c = 0; c < ncu; ++c) v.push_back((double)(h[c * 8 + k] - t0) * 0.01);
      std::sort(v.begin(), v.end());
      printf("  %-18s min %6.2f  median %6.2f  max %6.2f us\n", names[k], v.front(), v[v.size() / 2], v.back());
    }
  }
  return 0;
}
