// Probe: which lanes of a wave does ds_read_b64_tr_b16 serve together?  k_mlp_bwd128's activation tiles ([feature][sample], pitch 148
// halves) were rotated so that the four rows of a 16-lane group tile the 64 banks (prow), yet the PMC pass still reports 31-33 % of its
// LDS cycles as bank conflicts.  Three row maps, same instruction stream; run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE   (conflict share per kernel)
// and read the cycles per read of a lone wave per SIMD from stdout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
constexpr int PH = 148;
template <int MAP> __device__ inline int prow(int f) {
  if (MAP == 0) return f;                                                        // plain
  if (MAP == 1) return (f & ~31) | ((f & 3) << 3) | ((f >> 2) & 7);              // round 4: consecutive features 8 rows apart
  const int x = (f >> 2) & 7;                                                    // candidate: ... and feature + 4 four rows further
  return (f & ~31) | ((f & 3) << 3) | ((x & 1) << 2) | (x >> 1);
}
template <int MAP>
__global__ __launch_bounds__(256) void k_tr(long long* out, int rounds) {
  extern __shared__ __attribute__((aligned(16))) short lds[];
  for (int i = threadIdx.x; i < 128 * PH; i += 256) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, hf = lane >> 5;
  uint32_t addr[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {   // act_frag of mlp_bwd128.hip: k-step s & 7, row-block pair s >> 3
    const int ks = s & 7, rb = s >> 3;
    const int f = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);
    const int col = 32 * rb + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    addr[s] = (uint32_t)((prow<MAP>(f) * PH + col) * 2);
  }
  s4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    s4 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[s]) : "v"(addr[s]));
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int s = 0; s < 16; ++s) acc ^= v[s];
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc[0] == 12345 && acc[1] == 54321) out[1] = acc[2];
}
template <int MAP> void run(const char* name, long long* d) {
  const int rounds = 2000;
  hipFuncSetAttribute((const void*)k_tr<MAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_tr<MAP>, dim3(256), dim3(256), 150 * 1024, 0, d, rounds);   // one block per CU, one wave per SIMD
  hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s %7.1f ticks per 64-lane transposing read (16 in flight, lone wave per SIMD)\n", name, (double)h[0] / (rounds * 16.0));
}
int main() {
  long long* d; hipMalloc(&d, 16);
  run<0>("plain rows", d);
  run<1>("prow (round 4: +1 feature = +8 rows)", d);
  run<2>("prow2 (... and +4 features = +4 rows)", d);
  return 0;
}
