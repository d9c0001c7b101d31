// Probe 3: what is the unit of cost of a global float atomic? lanes vs cache lines vs instructions; int vs float; LDS atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
// PAT 0: random per lane; 1: wave-random line, lanes consecutive floats (256 B); 2: all lanes same address; 3: groups of 8 lanes share a 32B sector
// TYPE 0 float, 1 int, 2 float with return
template <int PAT, int TYPE>
__global__ void k(float* tab, size_t n, int per_thread, float* sink) {
  uint32_t s = fmix(blockIdx.x * 1024u + threadIdx.x), ws = fmix(blockIdx.x * 16u + (threadIdx.x >> 6) + 777u);
  int lane = threadIdx.x & 63; float acc = 0;
  for (int i = 0; i < per_thread; ++i) {
    s = fmix(s + i); ws = fmix(ws + i);
    size_t idx;
    if (PAT == 0) idx = s % n;
    if (PAT == 1) idx = ((ws % (n / 64)) * 64) + lane;
    if (PAT == 2) idx = (ws % n);
    if (PAT == 3) idx = ((fmix(ws + (lane >> 3)) % (n / 8)) * 8) + (lane & 7);
    if (TYPE == 0) unsafeAtomicAdd(tab + idx, 1.0f);
    if (TYPE == 1) atomicAdd((int*)tab + idx, 1);
    if (TYPE == 2) acc += unsafeAtomicAdd(tab + idx, 1.0f);
  }
  if (TYPE == 2 && acc == -1.f) *sink = acc;
}
__global__ void k_lds(float* out, int per_thread) {
  __shared__ float t[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) t[i] = 0;
  __syncthreads();
  uint32_t s = fmix(blockIdx.x * 1024u + threadIdx.x);
  for (int i = 0; i < per_thread; ++i) { s = fmix(s + i); atomicAdd(&t[s & 8191], 1.0f); }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t[5];
}
int main() {
  size_t n = 14229504; float* tab; float* sink; CK(hipMalloc(&tab, n * 4)); CK(hipMalloc(&sink, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = 2048, threads = 256, per = 128; double ops = (double)blocks * threads * per;
#define RUN(P, T, name) { float best = 1e9; for (int r = 0; r < 3; ++r) { CK(hipMemset(tab, 0, n * 4)); CK(hipEventRecord(e0)); \
    hipLaunchKernelGGL((k<P, T>), dim3(blocks), dim3(threads), 0, 0, tab, n, per, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } printf("%-50s %.3f ms  %.1f G lane-ops/s\n", name, best, ops / best / 1e6); }
  RUN(0, 0, "float random lanes");
  RUN(1, 0, "float 64 consecutive (one 256B span per wave)");
  RUN(3, 0, "float 8-lane groups share a 32B sector");
  RUN(2, 0, "float all lanes same address");
  RUN(0, 1, "int random lanes");
  RUN(1, 1, "int 64 consecutive");
  RUN(0, 2, "float random lanes WITH return");
  { float best = 1e9; for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, sink, per * 8);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-50s %.3f ms  %.1f G lane-ops/s\n", "LDS float atomicAdd random over 32KB", best, ops * 8 / best / 1e6); }
  return 0;
}
