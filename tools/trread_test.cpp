// trread_test.cpp -- pins the semantics of ds_read_b64_tr_b16 assumed by the bf16-KV MFMA attention kernel:
// within a 16-lane group, lane p supplies the address of 4 contiguous 16-bit elements = row p/4, columns (p%4)*4.. of a
// 4 x 16 block; lane i receives column i of that block (rows 0..3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
  __shared__ short lds[64 * 160];
  for (int i = threadIdx.x; i < 64 * 160; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, p = lane & 15;
  // group g reads rows g*4 .. g*4+3 (each `pitch` elements long), columns 32 .. 47
  const short* src = lds + (grp * 4 + (p >> 2)) * pitch + 32 + (p & 3) * 4;
  v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)src);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  int bad_total = 0;
  for (int pitch : {128, 144, 136}) {
    k<<<1, 64>>>(d, pitch);
    std::vector<short> h(256); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        const int grp = lane >> 4, i = lane & 15;
        const short expect = (short)((grp * 4 + j) * pitch + 32 + i);
        if (h[lane * 4 + j] != expect) { if (bad < 4) printf("pitch %d lane %d j %d got %d expect %d\n", pitch, lane, j, h[lane * 4 + j], expect); ++bad; }
      }
    printf("pitch %d: %d mismatches\n", pitch, bad); bad_total += bad;
  }
  return bad_total != 0;
}
