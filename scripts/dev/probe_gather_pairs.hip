// Probe: does a 4-byte gather get cheaper when adjacent lanes hit the same 64-byte chunk?  (hash-grid forward: the two
// x-neighbour corners of a cell share a chunk 15 times out of 16 -- issued from adjacent lanes of ONE instruction instead
// of from the same lane in two instructions.)  group = 1: every lane its own random entry; 2 / 4: lane groups share a
// random 64-byte chunk (consecutive entries).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ inline uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
template <int GROUP>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    uint32_t v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint32_t r = mix((tid / GROUP) * 977u + it * 8u + c + 12345u);
      uint32_t idx = ((r & mask) & ~(uint32_t)(GROUP - 1)) | (tid & (GROUP - 1));
      v[c] = tab[idx];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc ^= v[c];
  }
  if (acc == 0x1234567u) out[0] = acc;
}
int main() {
  for (int log2n = 19; log2n <= 23; log2n += 4) {
    uint32_t n = 1u << log2n;
    uint32_t* tab; uint32_t* out; hipMalloc(&tab, n * 4); hipMalloc(&out, 4); hipMemset(tab, 1, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096, iters = 16;   // 4096*256 lanes x 16 x 8 gathers = 134 M gathers (one encode pass)
    for (int g = 1; g <= 4; g *= 2) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        if (g == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, iters, out);
        if (g == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, iters, out);
        if (g == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, iters, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("table %2u MB, lanes per chunk %d: %.0f us for 134 M gathers (%.0f G lane-gathers/s)\n", n * 4 >> 20, g, best * 1e3,
             134.2 / best);
    }
    hipFree(tab); hipFree(out);
  }
  return 0;
}
