// Probe 2: L2-resident atomic rates (fp32 vs packed fp16), pair pattern, same-address contention.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ inline int xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// region_bytes: per-XCD private region size (L2-resident if <= ~2-3 MB); kind 0 = fp32 single, 1 = fp32 pair (2 adjacent), 2 = pk f16
template <int KIND>
__global__ void k(float* tab, size_t region_bytes, int per_thread, int shared_region) {
  int xcc = shared_region ? 0 : xcc_id();
  char* base = (char*)tab + (size_t)xcc * region_bytes;
  uint32_t s = fmix(blockIdx.x * 1024u + threadIdx.x);
  size_t n8 = region_bytes / 8, n4 = region_bytes / 4;
  for (int i = 0; i < per_thread; ++i) {
    s = fmix(s + i);
    if (KIND == 0) unsafeAtomicAdd((float*)base + s % n4, 1.0f);
    if (KIND == 1) { float* p = (float*)base + 2 * (s % n8); unsafeAtomicAdd(p, 1.0f); unsafeAtomicAdd(p + 1, 1.0f); }
    if (KIND == 2) { h2 v = {(_Float16)1.0f, (_Float16)1.0f}; __builtin_amdgcn_global_atomic_fadd_v2f16((h2*)base + s % n4, v); }
  }
}
int main() {
  size_t total = 64u << 20; float* tab; CK(hipMalloc(&tab, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 2048, threads = 256, per = 128;
  double ops = (double)blocks * threads * per;
  size_t regions[] = {32u << 10, 256u << 10, 1u << 20, 2u << 20, 4u << 20, 8u << 20};
  for (int kind = 0; kind < 3; ++kind) for (int shared = 0; shared < 2; ++shared) for (size_t r : regions) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(tab, 0, total));
      CK(hipEventRecord(e0));
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, tab, r, per, shared);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, tab, r, per, shared);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, tab, r, per, shared);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("kind %d (%s) %s region %5zu KB: %.3f ms  %.1f G lane-ops/s\n", kind, kind == 0 ? "f32" : kind == 1 ? "f32 pair" : "pk_f16",
           shared ? "ALL XCDs SAME region" : "per-XCD private", r >> 10, best, ops / best / 1e6);
  }
  return 0;
}
