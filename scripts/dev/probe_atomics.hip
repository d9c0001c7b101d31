// Probe: fp32 atomic-add throughput on MI355X by scope and XCD locality; XCC_ID placement census.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ inline int xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 0xf; }

// mode 0: agent scope random over whole table; mode 1: workgroup scope, address region = own XCD slice; mode 2: agent scope own slice
template <int MODE>
__global__ void k_atomics(float* tab, size_t n_entries, int per_thread, int* census) {
  int xcc = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&census[xcc], 1);
  uint32_t s = fmix(blockIdx.x * 1024u + threadIdx.x);
  size_t slice = n_entries / 8;
  for (int i = 0; i < per_thread; ++i) {
    s = fmix(s + i);
    size_t idx = (MODE == 0) ? (s % n_entries) : (xcc * slice + s % slice);
    if (MODE == 1) __hip_atomic_fetch_add(tab + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(tab + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_sum(const float* tab, size_t n, double* out) {
  double a = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += tab[i];
  atomicAdd(out, a);
}

int main() {
  size_t n = 14229504;  // fp32 grid gradient entries
  float* tab; int* census; double* sum;
  CK(hipMalloc(&tab, n * 4)); CK(hipMalloc(&census, 64)); CK(hipMalloc(&sum, 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 2048, threads = 256, per = 256;
  double total = (double)blocks * threads * per;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(tab, 0, n * 4)); CK(hipMemset(census, 0, 64)); CK(hipMemset(sum, 0, 8));
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(blocks), dim3(threads), 0, 0, tab, n, per, census);
      if (mode == 1) hipLaunchKernelGGL(k_atomics<1>, dim3(blocks), dim3(threads), 0, 0, tab, n, per, census);
      if (mode == 2) hipLaunchKernelGGL(k_atomics<2>, dim3(blocks), dim3(threads), 0, 0, tab, n, per, census);
      hipEventRecord(e1); CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, tab, n, sum); CK(hipDeviceSynchronize());
      double hs; int hc[16]; CK(hipMemcpy(&hs, sum, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, census, 64, hipMemcpyDeviceToHost));
      printf("mode %d rep %d: %.3f ms  %.1f G atomics/s  sum %.0f (expect %.0f)  census", mode, rep, ms, total / ms / 1e6, hs, total);
      for (int i = 0; i < 8; ++i) printf(" %d", hc[i]);
      printf("\n");
    }
  }
  return 0;
}
