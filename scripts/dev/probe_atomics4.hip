// Probe 4: request rate by atomic type with the encode_bwd access pattern (16 adjacent lanes = one cell: 4 random 16-byte
// spans of [x-pair][2 features]); and effect of waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
// TYPE 0 f32, 1 i32, 2 u64, 3 f64
template <int TYPE>
__global__ void k(void* tab, size_t n_entries, int per_thread) {
  int lane = threadIdx.x & 63;
  uint32_t gs = fmix(blockIdx.x * 4096u + (threadIdx.x >> 2) * 7u + 13u);  // one random span per 4 lanes
  for (int i = 0; i < per_thread; ++i) {
    gs = fmix(gs + i);
    size_t entry = ((size_t)(gs % (n_entries / 2)) * 2) + ((lane >> 1) & 1);   // x-pair: aligned entry pair
    size_t idx = entry * 2 + (lane & 1);                                       // feature
    if (TYPE == 0) unsafeAtomicAdd((float*)tab + idx, 1.0f);
    if (TYPE == 1) atomicAdd((int*)tab + idx, 1);
    if (TYPE == 2) atomicAdd((unsigned long long*)tab + idx, 1ull);
    if (TYPE == 3) unsafeAtomicAdd((double*)tab + idx, 1.0);
  }
}
int main() {
  size_t n_entries = 7114752; void* tab; CK(hipMalloc(&tab, n_entries * 2 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"f32", "i32", "u64", "f64"};
  for (int threads : {256, 1024}) for (int blocks : {512, 2048, 8192}) for (int t = 0; t < 4; ++t) {
    int per = 128; double ops = (double)blocks * threads * per; float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipMemset(tab, 0, n_entries * 16)); CK(hipEventRecord(e0));
      if (t == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per);
      if (t == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per);
      if (t == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per);
      if (t == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, tab, n_entries, per);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%s threads %4d blocks %5d: %.3f ms  %.1f G lane-ops/s  %.1f G spans/s\n", names[t], threads, blocks, best, ops / best / 1e6, ops / 4 / best / 1e6);
  }
  return 0;
}
