// Fused Adam + grad unscale + inf-skip + fp16 shadow refresh + grad zeroing, and the GradScaler state update.
// Replaces torch.optim.Adam (scripts/train.py:50-63: lr 5e-3, betas (0.9,0.99), eps 1e-15, weight_decay 1e-6
// on the MLP group only) and torch.cuda.amp.GradScaler step/update (autolabel/trainer.py:45-48).
// Spec: oracle/nerf_oracle.py:adam_update.  HBM-bound streaming kernel over the flat parameter buffer
// [grid table | MLP weights]: 16 B read + 18 B written per parameter.
#include "common.h"
#include <math.h>

// state (device, int32/float32 mixed view):
//   si[0] = optimizer step count t (counts applied steps only)   si[1] = growth tracker   si[2] = found_inf flag
//   sf[0] = loss scale
// consts (written by k_adam_prepare): c[0] = skip (0/1), c[1] = lr / bc1, c[2] = 1/sqrt(bc2), c[3] = 1/scale_used
struct AdamHyper { float lr, beta1, beta2, eps, wd_net, growth, backoff; int growth_interval; };

__global__ void k_adam_prepare(int* si, float* sf, float* c, AdamHyper h) {
  int found = si[2];
  float scale = sf[0];
  c[3] = 1.0f / scale;
  if (found) {
    c[0] = 1.f; c[1] = 0.f; c[2] = 1.f;
    sf[0] = scale * h.backoff; si[1] = 0;
  } else {
    int t = si[0] + 1; si[0] = t;
    double bc1 = 1.0 - pow((double)h.beta1, (double)t), bc2 = 1.0 - pow((double)h.beta2, (double)t);
    c[0] = 0.f; c[1] = (float)((double)h.lr / bc1); c[2] = (float)(1.0 / sqrt(bc2));
    int tr = si[1] + 1;
    if (tr >= h.growth_interval) { sf[0] = scale * h.growth; tr = 0; }
    si[1] = tr;
  }
  si[2] = 0;
}

__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       h16* __restrict__ table16, size_t n_grid, size_t n_total, const float* __restrict__ c, AdamHyper h) {
  const bool skip = c[0] != 0.f;
  const float step_size = c[1], inv_sqrt_bc2 = c[2], inv_scale = c[3];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_total; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    g[i] = 0.f;
    if (skip) continue;
    float pi = p[i];
    gi *= inv_scale;
    if (i >= n_grid) gi += h.wd_net * pi;
    float mi = h.beta1 * m[i] + (1.f - h.beta1) * gi;
    float vi = h.beta2 * v[i] + (1.f - h.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) * inv_sqrt_bc2 + h.eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
    if (i < n_grid) table16[i] = (h16)pi;
  }
}

extern "C" int aln_adam_step(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid,
                             int64_t n_total, int32_t* state_i, float* state_f, float* consts, float lr, float beta1,
                             float beta2, float eps, float wd_net, float growth, float backoff, int32_t growth_interval,
                             void* stream) {
  ALN_REQUIRE(params && grads && m && v && state_i && state_f && consts, "adam: NULL pointer");
  ALN_REQUIRE(n_grid == 0 || table_f16, "adam: fp16 table shadow missing");
  AdamHyper h{lr, beta1, beta2, eps, wd_net, growth, backoff, growth_interval};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(1), 0, s, state_i, state_f, consts, h);
  ALN_CHECK_LAUNCH("adam_prepare");
  hipLaunchKernelGGL(k_adam, dim3(aln_grid_for(n_total, 256, 256 * 16)), dim3(256), 0, s, params, grads, m, v, (h16*)table_f16,
                     (size_t)n_grid, (size_t)n_total, consts, h);
  ALN_CHECK_LAUNCH("adam");
  return 0;
}

// fp16 shadow of the grid table from the fp32 master (initialisation / checkpoint load)
__global__ void k_cast_f16(const float* __restrict__ src, h16* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (h16)src[i];
}
extern "C" int aln_cast_f16(const float* src, void* dst, int64_t n, void* stream) {
  ALN_REQUIRE(src && dst, "cast_f16: NULL pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_cast_f16, dim3(aln_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, src, (h16*)dst, (size_t)n);
  ALN_CHECK_LAUNCH("cast_f16");
  return 0;
}
