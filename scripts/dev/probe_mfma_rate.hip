// Probe: issue rate of v_mfma_f32_32x32x16_f16 on gfx950 in the shapes the fused MLP backward uses.
//   mode 0: 4 independent accumulators, register operands only
//   mode 1: + one ds_read_b128 A-fragment prefetch per MFMA (double-buffered, as chain_layer does)
//   mode 2: as 1 with ds_read2_b64 (two 8-byte pieces 16 B apart)
// build twice: default (AGPR accumulators allowed) and with -mllvm -amdgpu-mfma-vgpr-form=1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[32 * 1024];
  for (int i = threadIdx.x; i < 32 * 1024; i += blockDim.x) lds[i] = (_Float16)(i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  h16x8 b; for (int j = 0; j < 8; ++j) b[j] = (_Float16)(lane & 3);
  h16x8 a[2][4];
  auto frag = [&](int m, int ks) -> h16x8 {
    const _Float16* p = lds + (32 * m + (lane & 31)) * 136 + 16 * ks + 8 * (lane >> 5);
    if (MODE == 1) return *(const h16x8*)p;
    h16x4 x = *(const h16x4*)p, y = *(const h16x4*)(p + 8 - 4 * (lane >> 5));
    h16x8 r; for (int j = 0; j < 4; ++j) { r[j] = x[j]; r[4 + j] = y[j]; } return r;
  };
  for (int m = 0; m < 4; ++m) a[0][m] = a[1][m] = b;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (MODE != 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) a[(ks + 1) & 1][m] = frag(m, (ks + 1) & 7);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][m], b, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = 0; for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  long long* o; float* sink; hipMalloc(&o, 8); hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 2000;
  for (int waves = 4; waves <= 8; waves += 4)
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, o, sink, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, o, sink, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, o, sink, iters);
      }
      hipDeviceSynchronize();
      long long t; hipMemcpy(&t, o, 8, hipMemcpyDeviceToHost);
      printf("waves/CU %d mode %d: %.1f cycles per MFMA per wave\n", waves, mode, (double)t / (iters * 32));
    }
  return 0;
}
