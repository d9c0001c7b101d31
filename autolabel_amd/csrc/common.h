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

// tcnn SphericalHarmonics degree 4 on d01 in [0,1] (models.py:205-207) -- oracle: sh4_encode
__device__ inline void sh4(float x, float y, float z, float* o) {
  float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy; o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f; o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2); o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// the remap of models.py:205 ((d+1)/2, which tcnn maps back with 2x-1) followed by SH deg 4
__device__ inline void sh4_of_dir(const float* d, float* sh) {
  float v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { float d01 = (d[k] + 1.0f) / 2.0f; v[k] = d01 * 2.0f - 1.0f; }
  sh4(v[0], v[1], v[2], sh);
}

