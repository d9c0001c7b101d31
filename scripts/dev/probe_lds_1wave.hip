// Probe (round 4): what one wave per SIMD gets out of the LDS and the matrix pipe on gfx950.
//   mfma      : 8 independent v_mfma_f32_32x32x16_f16 back to back (ticks per MFMA, and the wall clock -> shader clock)
//   read K    : 16 LDS reads of kind K in flight, then wait (ticks per read instruction), 4 waves per CU and 8 waves per CU
//   mix       : 8 MFMAs + 8 tr reads + 8 b64 reads per step, software pipelined (phase C of k_mlp_bwd128)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4v* lds_s16x4_ptr;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LV(T) __attribute__((address_space(3))) T
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int KIND>   // 0 b64 row-per-lane (pitch 148), 1 tr_b64 (pitch 148), 2 b128 row-per-lane (pitch 72), 3 b128 lane-linear, 4 ds_read2_b64 (two pieces 16 B apart)
__global__ __launch_bounds__(512) void k_read(long long* out, float* sink, int iters) {
  __attribute__((address_space(3))) h16* t = (__attribute__((address_space(3))) h16*)smem;
  for (int i = threadIdx.x; i < 64 * 1024; i += blockDim.x) t[i] = (h16)(i & 15);
  __syncthreads();
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  float s = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[16];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (KIND == 0) { u32x2 x = *(const LV(u32x2)*)(t + c * 148 + 16 * j + 4 * hf + 8 * (it & 1)); v[j] = (u32x4){x.x, x.y, 0, 0}; }
      if (KIND == 1) {
        const int row = 4 * hf + ((lane & 15) >> 2) + 8 * (j & 7), col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 32 * (j >> 3);
        s16x4v x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(t + row * 148 + col + 64 * (it & 1)));
        v[j] = (u32x4){(uint32_t)x[0], (uint32_t)x[1], 0, 0};
      }
      if (KIND == 2) v[j] = *(const LV(u32x4)*)(t + (c + 32 * (j & 3)) * 72 + 16 * (j >> 2) + 8 * hf);
      if (KIND == 3) v[j] = *(const LV(u32x4)*)(t + (j * 64 + lane) * 8);
      if (KIND == 4) {
        const __attribute__((address_space(3))) h16* p = t + c * 148 + 16 * j + 4 * hf;
        u32x2 x = *(const LV(u32x2)*)p, y = *(const LV(u32x2)*)(p + 8);
        v[j] = (u32x4){x.x, x.y, y.x, y.y};
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += __builtin_bit_cast(float, v[j].x ^ v[j].z);
  }
  long long t1 = clock64();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MIX>   // 0: MFMAs only; 1: + 8 tr reads and 8 b64 reads per 8 MFMAs, one step ahead; 2: only the tr reads; 3: only the b64 reads
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mix(long long* out, float* sink, int iters) {
  __attribute__((address_space(3))) h16* t = (__attribute__((address_space(3))) h16*)smem;
  for (int i = threadIdx.x; i < 64 * 1024; i += blockDim.x) t[i] = (h16)((i & 15) * 0.01f);
  __syncthreads();
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  f32x16 acc[8];
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  h16x8 a[2][4], b[2][4], w;
  for (int j = 0; j < 8; ++j) w[j] = (h16)(0.01f * (lane & 3));
  for (int m = 0; m < 4; ++m) a[0][m] = a[1][m] = b[0][m] = b[1][m] = w;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (MIX == 1 || MIX == 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2), col = 32 * m + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(t + row * 148 + col));
          s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(t + (row + 8) * 148 + col));
          union { struct { s16x4v l, h; } s; h16x8 v; } u; u.s.l = lo; u.s.h = hi; a[(ks + 1) & 1][m] = u.v;
        }
      }
      if (MIX == 1 || MIX == 3) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const __attribute__((address_space(3))) h16* p0 = t + 128 * 148 + (32 * m + c) * 148 + 16 * ks + 4 * hf;
          const __attribute__((address_space(3))) h16* p1 = p0 + 8;
          asm volatile("" : "+v"(p1));
          union { struct { u32x2 x, y; } s; h16x8 v; } u; u.s.x = *(const LV(u32x2)*)p0; u.s.y = *(const LV(u32x2)*)p1; b[(ks + 1) & 1][m] = u.v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][m], w, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[4 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b[ks & 1][m], acc[4 + m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = 0; for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  long long* o; float* sink; hipMalloc(&o, 8); hipMalloc(&sink, 256 * 512 * 4);
  const int lds = 128 * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* kn[] = {"ds_read_b64 rows", "ds_read_b64_tr_b16", "ds_read_b128 rows", "ds_read_b128 linear", "2 x b64 (16 B apart)"};
#define RUN(K, threads, iters, per, label)                                                     \
  do {                                                                                        \
    hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, lds);     \
    hipLaunchKernelGGL(K, dim3(256), dim3(threads), lds, 0, o, sink, iters);                  \
    hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(threads), lds, 0, o, sink, iters); hipEventRecord(e1); \
    hipDeviceSynchronize(); long long t; hipMemcpy(&t, o, 8, hipMemcpyDeviceToHost); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%-28s %d waves/CU: %7.1f ticks per %s   (kernel %.1f us -> %.2f G ticks/s)\n", label, threads / 64, (double)t / ((double)iters * per), \
           #per, ms * 1e3, (double)t / (ms * 1e6));                                           \
  } while (0)
  for (int threads = 256; threads <= 512; threads += 256) {
    RUN(k_read<0>, threads, 2000, 16, kn[0]); RUN(k_read<1>, threads, 2000, 16, kn[1]); RUN(k_read<2>, threads, 2000, 16, kn[2]);
    RUN(k_read<3>, threads, 2000, 16, kn[3]); RUN(k_read<4>, threads, 2000, 16, kn[4]);
  }
  RUN(k_mix<0>, 256, 500, 64, "mfma only (8 acc)"); RUN(k_mix<1>, 256, 500, 64, "mfma + 8 tr + 8 b64 / step");
  RUN(k_mix<2>, 256, 500, 64, "mfma + 8 tr / step"); RUN(k_mix<3>, 256, 500, 64, "mfma + 8 b64 / step");
  return 0;
}
