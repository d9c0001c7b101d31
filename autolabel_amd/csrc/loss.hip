// Per-ray loss and its gradient, reduced on device (no host sync).
// Replaces autolabel/trainer.py:72-92 (SimpleTrainer.train_step): rgb MSE + depth L1 over valid depth +
// feature L1 + cross-entropy over labelled rays.  Spec: oracle/nerf_oracle.py:loss_fn.
#include "common.h"
#include <math.h>

#define DEPTH_EPSILON 0.01f

__global__ void k_loss_counts(const float* __restrict__ gt_depth, const int* __restrict__ gt_sem, int N, int* __restrict__ counts) {
  int nd = 0, ns = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    nd += gt_depth[i] > DEPTH_EPSILON; ns += gt_sem[i] >= 0;
  }
  nd = (int)wave_sum((float)nd); ns = (int)wave_sum((float)ns);
  if ((threadIdx.x & 63) == 0) { if (nd) atomicAdd(counts, nd); if (ns) atomicAdd(counts + 1, ns); }
}

struct LossArgs {
  const float* image; const float* depth; const float* semantic; const float* features;
  const float* gt_rgb; const float* gt_depth; const int* gt_sem; const float* gt_feat;
  int N, C, D, Cf; float w_rgb, w_depth, w_sem, w_feat;
  const int* counts; const float* loss_scale;
  float* g_image; float* g_depth; float* g_sem; float* g_feat; float* terms;  // terms[5]: rgb, depth, feature, semantic, total
};

// one wavefront per ray (4 per block): every ray is a short chain of dependent loads, so the kernel is latency-bound and
// wants as many rays in flight as the chip holds; the block folds its four partial sums before the 5 same-address atomics
__global__ __launch_bounds__(256) void k_loss(LossArgs a) {
  __shared__ float part[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float scale = a.loss_scale ? *a.loss_scale : 1.f;
  const int nd = a.counts[0], ns = a.counts[1];
  float t_rgb = 0, t_depth = 0, t_feat = 0, t_sem = 0;
  for (int ray = blockIdx.x * 4 + wave; ray < a.N; ray += gridDim.x * 4) {
    if (lane < 3) {
      float diff = a.image[3 * (size_t)ray + lane] - a.gt_rgb[3 * (size_t)ray + lane];
      t_rgb += diff * diff;
      a.g_image[3 * (size_t)ray + lane] = scale * a.w_rgb * 2.f * diff / (3.f * a.N);
    }
    if (lane == 0) {
      float g = 0.f, gd = a.gt_depth[ray];
      if (gd > DEPTH_EPSILON) {
        float diff = a.depth[ray] - gd;
        t_depth += fabsf(diff);
        g = scale * a.w_depth * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / (float)nd;
      }
      a.g_depth[ray] = g;
    }
    if (a.g_feat) {
      for (int d = lane; d < a.D; d += 64) {
        float g = 0.f;
        if (a.gt_feat && d < a.Cf) {
          float diff = a.features[(size_t)ray * a.D + d] - a.gt_feat[(size_t)ray * a.Cf + d];
          t_feat += fabsf(diff);
          g = scale * a.w_feat * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / ((float)a.N * a.Cf);
        }
        a.g_feat[(size_t)ray * a.D + d] = g;
      }
    }
    if (a.g_sem) {
      int label = a.gt_sem[ray];
      if (label >= 0) {
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, a.semantic[(size_t)ray * a.C + c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f;
        for (int c = lane; c < a.C; c += 64) se += expf(a.semantic[(size_t)ray * a.C + c] - mx);
        se = wave_sum(se);
        for (int c = lane; c < a.C; c += 64) {
          float l = a.semantic[(size_t)ray * a.C + c];
          float pr = expf(l - mx) / se;
          if (c == label) t_sem += -(l - mx - logf(se));
          a.g_sem[(size_t)ray * a.C + c] = scale * a.w_sem * (pr - (c == label ? 1.f : 0.f)) / (float)ns;
        }
      } else {
        for (int c = lane; c < a.C; c += 64) a.g_sem[(size_t)ray * a.C + c] = 0.f;
      }
    }
  }
  t_rgb = wave_sum(t_rgb); t_depth = wave_sum(t_depth); t_feat = wave_sum(t_feat); t_sem = wave_sum(t_sem);
  if (lane == 0) { part[wave][0] = t_rgb; part[wave][1] = t_depth; part[wave][2] = t_feat; part[wave][3] = t_sem; }
  __syncthreads();
  if (threadIdx.x == 0 && a.terms) {
    t_rgb = part[0][0] + part[1][0] + part[2][0] + part[3][0]; t_depth = part[0][1] + part[1][1] + part[2][1] + part[3][1];
    t_feat = part[0][2] + part[1][2] + part[2][2] + part[3][2]; t_sem = part[0][3] + part[1][3] + part[2][3] + part[3][3];
    float r = t_rgb / (3.f * a.N), d = nd ? t_depth / nd : 0.f, f = (a.gt_feat && a.Cf) ? t_feat / ((float)a.N * a.Cf) : 0.f,
          s = ns ? t_sem / ns : 0.f;
    atomicAdd(a.terms, r); atomicAdd(a.terms + 1, d); atomicAdd(a.terms + 2, f); atomicAdd(a.terms + 3, s);
    atomicAdd(a.terms + 4, a.w_rgb * r + a.w_depth * d + a.w_feat * f + a.w_sem * s);
  }
}

extern "C" int aln_loss_fwd_bwd(const float* image, const float* depth, const float* semantic, const float* features,
                                const float* gt_rgb, const float* gt_depth, const int32_t* gt_sem, const float* gt_feat,
                                int32_t N, int32_t C, int32_t D, int32_t Cf, float w_rgb, float w_depth, float w_sem,
                                float w_feat, const float* loss_scale, int32_t* counts, float* g_image, float* g_depth,
                                float* g_sem, float* g_feat, float* terms, void* stream) {
  ALN_REQUIRE(image && depth && gt_rgb && gt_depth && gt_sem && counts && g_image && g_depth, "loss: NULL pointer");
  ALN_REQUIRE(!g_sem || semantic, "loss: semantic output missing");
  ALN_REQUIRE(!gt_feat || (features && g_feat && Cf <= D), "loss: feature buffers missing or Cf > D");
  if (N <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipMemsetAsync(counts, 0, 2 * sizeof(int), s);
  if (terms) hipMemsetAsync(terms, 0, 5 * sizeof(float), s);
  hipLaunchKernelGGL(k_loss_counts, dim3(aln_grid_for(N, 256, 64)), dim3(256), 0, s, gt_depth, gt_sem, N, counts);
  ALN_CHECK_LAUNCH("loss_counts");
  LossArgs a{image, depth, semantic, features, gt_rgb, gt_depth, gt_sem, gt_feat, N, C, D, Cf, w_rgb, w_depth, w_sem, w_feat,
             counts, loss_scale, g_image, g_depth, g_sem, g_feat, terms};
  int nb = (N + 3) / 4;
  // (the 5 loss-term atomics of every block hit one cache line and serialize in L2: 256 blocks, not one per 4 rays)
  hipLaunchKernelGGL(k_loss, dim3(nb < 256 ? nb : 256), dim3(256), 0, s, a);
  ALN_CHECK_LAUNCH("loss");
  return 0;
}
