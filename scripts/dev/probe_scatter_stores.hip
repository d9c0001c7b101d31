// Probe: rate of scattered small stores (the first pass of a "bin the (index, value) records, then accumulate each bin in
// LDS" alternative to the atomic scatter of k_encode_bwd).  mode 0: every lane stores 8 bytes to a random 8-byte slot of a
// 1 GB buffer; mode 1: 12 bytes (3 dwords) to a random 16-byte slot; mode 2: records appended to 1024 bin queues, each
// wave hitting 64 different bins per instruction (what an unsorted wave of records looks like).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ inline uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* buf, size_t nslots, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    uint32_t r = mix(tid * 977u + it * 131u + 7u);
    if (MODE == 0) { uint2 v = make_uint2(r, it); *(uint2*)(buf + 2 * (size_t)(r % (uint32_t)(nslots / 2))) = v; }
    if (MODE == 1) { uint32_t* p = buf + 4 * (size_t)(r % (uint32_t)(nslots / 4)); p[0] = r; p[1] = it; p[2] = tid; }
    if (MODE == 2) {   // bin = random of 1024, slot within the bin advances with (block, it): 12-byte records, bins 1 MB apart
      uint32_t bin = r & 1023u; size_t slot = ((size_t)blockIdx.x * iters + it) * 4 + (threadIdx.x >> 6);
      uint32_t* p = buf + (size_t)bin * (nslots / 1024) + (slot % (nslots / 1024 / 4)) * 3;
      p[0] = r; p[1] = it; p[2] = tid;
    }
  }
}
int main() {
  const size_t bytes = 1ull << 30; uint32_t* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 4096, iters = 128;   // 4096 * 256 * 128 = 134 M records = one training step's (corner, sample, level) updates
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 4, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("mode %d: %.2f ms for 134 M records (%.1f G records/s)\n", mode, best, 134.2 / best);
  }
  return 0;
}
