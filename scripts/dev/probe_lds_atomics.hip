// Probe: LDS atomic throughput by type and address pattern (what bounds phase 2 of the binned hash-grid backward).
// OP 0 ds_add_f32, 1 ds_add_u32, 2 ds_add_u64, 3 ds_pk_add_f16 (via builtin), 4 plain ds_write_b32 (reference), 5 ds_add_rtn_u32
// PAT 0 random over 64 KB, 1 conflict-free (lane-distinct banks, random row), 2 random but 2 adjacent floats per lane (the record pattern)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
template <int OP, int PAT>
__global__ void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float t[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) t[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t s = fmix(blockIdx.x * 4096u + threadIdx.x);
  unsigned long long* t64 = (unsigned long long*)t;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    s = fmix(s + i);
    uint32_t a;
    if (PAT == 0) a = s & 16383u;
    if (PAT == 1) a = ((__builtin_amdgcn_readfirstlane(s) & 255u) << 6) + lane;
    if (PAT == 2) a = (s & 8191u) << 1;
    if (OP == 0) { unsafeAtomicAdd(&t[a], 1.0f); if (PAT == 2) unsafeAtomicAdd(&t[a + 1], 1.0f); }
    if (OP == 1) { atomicAdd((uint32_t*)&t[a], 1u); if (PAT == 2) atomicAdd((uint32_t*)&t[a + 1], 1u); }
    if (OP == 2) { atomicAdd(&t64[a >> 1], 1ull); }
    if (OP == 4) { t[a] = (float)i; if (PAT == 2) t[a + 1] = (float)i; }
    if (OP == 5) { acc += atomicAdd((uint32_t*)&t[a], 1u); }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = t[5] + acc;
}
int main() {
  float* sink; CK(hipMalloc(&sink, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048;
#define RUN(OP, PAT, threads, blocks, name) { float best = 1e9; for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); \
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), 0, 0, sink, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
    double ops = (double)blocks * threads * iters * ((PAT == 2 && OP != 2) ? 2 : 1); \
    printf("%-34s threads %4d blocks %5d: %7.3f ms %7.1f G lane-ops/s  %.2f lanes/clk/CU\n", name, threads, blocks, best, ops / best / 1e6, ops / best / 1e6 / 256 / 2.4); }
  for (int threads : {256, 512, 1024}) {
    int blocks = 512 * 1024 / threads * 2;
    RUN(0, 0, threads, blocks, "ds_add_f32 random");
    RUN(0, 1, threads, blocks, "ds_add_f32 conflict-free");
    RUN(0, 2, threads, blocks, "ds_add_f32 random pair");
    RUN(1, 0, threads, blocks, "ds_add_u32 random");
    RUN(1, 1, threads, blocks, "ds_add_u32 conflict-free");
    RUN(2, 2, threads, blocks, "ds_add_u64 random (pair slot)");
    RUN(5, 0, threads, blocks, "ds_add_rtn_u32 random");
    RUN(4, 0, threads, blocks, "ds_write_b32 random");
  }
  return 0;
}
