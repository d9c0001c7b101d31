// Shared device/host helpers for libautolabel_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/autolabel_hip.h"

#define ALN_WAVE 64

void aln_set_error(const char* fmt, ...);

#define ALN_CHECK_LAUNCH(name)                                                  \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      aln_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return -2;                                                                \
    }                                                                           \
  } while (0)

#define ALN_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      aln_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- counter RNG, bit-identical to oracle/nerf_oracle.py:rand_u32
__host__ __device__ inline uint32_t aln_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ inline uint32_t aln_rand_key(uint32_t seed, uint32_t stream, uint32_t step) {
  uint32_t key = aln_fmix32(seed + 0x9E3779B9u * (stream + 1u));
  return aln_fmix32(key ^ step);
}
__host__ __device__ inline uint32_t aln_rand_u32(uint32_t key, uint32_t idx) {
  uint32_t r = aln_fmix32(idx * 0x9E3779B1u + key);
  return aln_fmix32(r ^ 0x68E31DA4u);
}
__host__ __device__ inline float aln_rand_uniform(uint32_t key, uint32_t idx) {
  return (float)(aln_rand_u32(key, idx) >> 8) * (1.0f / 16777216.0f);
}
enum { ALN_STREAM_FRAME = 0, ALN_STREAM_PIXEL, ALN_STREAM_JX, ALN_STREAM_JY, ALN_STREAM_PERTURB, ALN_STREAM_PDF,
       ALN_STREAM_CLASS };

// sample position: clamp(o + d*z) evaluated unfused (oracle: torch ops)
__device__ inline void aln_sample_xyz(const float* __restrict__ o, const float* __restrict__ d, float z, float bound,
                                      float* x) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float p = __fadd_rn(o[k], __fmul_rn(d[k], z));
    x[k] = fminf(fmaxf(p, -bound), bound);
  }
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

static inline int aln_grid_for(int64_t work, int block, int max_blocks = 256 * 8) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
