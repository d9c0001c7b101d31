// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds element e at halfword index e (value = e).
// Each lane supplies its own byte address; print which elements each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x;
  // mode 0: lane address = lane * 8 bytes (contiguous).  mode 1: address = row-major [k][64] tile: row = (lane&15)>>2 ... see host print
  uint32_t addr;
  if (mode == 0) addr = lane * 8;
  else { int g = lane >> 4, l = lane & 15; int row = l >> 2, colblk = l & 3; addr = (uint32_t)((row * 64 + g * 16 + colblk * 4) * 2); }
  addr += (uint32_t)(uintptr_t)lds;  // LDS base offset (generic->local truncation works for static LDS at 0)
  s4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4); int h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode); hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
