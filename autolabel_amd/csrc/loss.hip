// Per-ray loss and its gradient, reduced on device (no host sync).
// Replaces autolabel/trainer.py:72-92 (SimpleTrainer.train_step): rgb MSE + depth L1 over valid depth +
// feature L1 + cross-entropy over labelled rays.  Spec: oracle/nerf_oracle.py:loss_fn.
#include "common.h"
#include <math.h>

#define DEPTH_EPSILON 0.01f

struct LossArgs {
  const float* image; const float* depth; const float* semantic; const float* features;
  const float* gt_rgb; const float* gt_depth; const int* gt_sem; const float* gt_feat;
  int N, C, D, Cf; float w_rgb, w_depth, w_sem, w_feat;
  int* counts; const float* loss_scale;
  float* g_image; float* g_depth; float* g_sem; float* g_feat; float* terms;  // terms[5]: rgb, depth, feature, semantic, total; then scratch
};
#define LOSS_MAX_BLOCKS 256

// ONE launch.  (a) every block counts the rays with a valid depth / a label itself (N is a batch: 4096 ... 32768 rays = a few
// hundred KB of L2 reads per block; integer sums, so every block holds the same two numbers) -- no separate count kernel, no
// zero-fill of counters; (b) one wavefront per ray: every ray is a short chain of dependent loads, so the kernel is latency-bound
// and wants as many rays in flight as the chip holds -- round 6: SIXTEEN waves per block (four made every wave of a 4096-ray batch
// walk four rays one after the other behind the count loop: 20 us; now one ray per wave and one trip of the count loop);
// (c) the block's partial loss sums go to its row of the scratch behind `terms`, and the LAST block to arrive (ticket in counts[2],
// self-resetting) adds the rows up in block order: the reported loss terms are bit-reproducible and nothing needs to be zeroed before
// the launch.
#define LOSS_WAVES 16
__global__ __launch_bounds__(64 * LOSS_WAVES) void k_loss(LossArgs a) {
  __shared__ float part[LOSS_WAVES][4];
  __shared__ int cnt_s[LOSS_WAVES][2];
  __shared__ int last_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float scale = a.loss_scale ? *a.loss_scale : 1.f;
  // Everything a ray reads, requested unconditionally (clamped lanes, values dropped where they do not apply): with a load per
  // `if (lane < 3)` / `if (lane == 0)` / label branch hipcc waited for each in turn -- ten HBM round trips per ray in a kernel that is
  // nothing but one wave's chain of them (round 6).  The first ray's request goes out BEFORE the count loop below: one trip for both.
  struct RayIn { float im, gt, gd, dp, sem, f, gf; int label; };
  const bool feat_gt = a.g_feat && a.gt_feat && a.Cf > 0;
  auto fetch = [&](int ray) {
    RayIn q{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, -1};
    ray = min(ray, a.N - 1);   // (past the batch: a valid ray, never used -- a branch here would put a wait behind it)
    const int l3 = min(lane, 2);
    q.im = a.image[3 * (size_t)ray + l3]; q.gt = a.gt_rgb[3 * (size_t)ray + l3];
    q.gd = a.gt_depth[ray]; q.dp = a.depth[ray];
    if (a.g_sem) { q.label = a.gt_sem[ray]; q.sem = a.semantic[(size_t)ray * a.C + min(lane, a.C - 1)]; }
    if (feat_gt) { const int d = min(lane, a.Cf - 1); q.f = a.features[(size_t)ray * a.D + d]; q.gf = a.gt_feat[(size_t)ray * a.Cf + d]; }
    return q;
  };
  RayIn cur = fetch(blockIdx.x * LOSS_WAVES + wave);
  int nd = 0, ns = 0;
  {   // 16-byte loads over the aligned body of the two arrays, the tail element-wise
    const int n4 = (((uintptr_t)a.gt_depth | (uintptr_t)a.gt_sem) & 15) == 0 ? a.N / 4 : 0;
    for (int i = threadIdx.x; i < n4; i += 64 * LOSS_WAVES) {
      const float4 d4 = ((const float4*)a.gt_depth)[i]; const int4 s4 = ((const int4*)a.gt_sem)[i];
      nd += (d4.x > DEPTH_EPSILON) + (d4.y > DEPTH_EPSILON) + (d4.z > DEPTH_EPSILON) + (d4.w > DEPTH_EPSILON);
      ns += (s4.x >= 0) + (s4.y >= 0) + (s4.z >= 0) + (s4.w >= 0);
    }
    for (int i = 4 * n4 + threadIdx.x; i < a.N; i += 64 * LOSS_WAVES) { nd += a.gt_depth[i] > DEPTH_EPSILON; ns += a.gt_sem[i] >= 0; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { nd += __shfl_xor(nd, o); ns += __shfl_xor(ns, o); }
  if (lane == 0) { cnt_s[wave][0] = nd; cnt_s[wave][1] = ns; }
  __syncthreads();
  nd = ns = 0;
#pragma unroll
  for (int w = 0; w < LOSS_WAVES; ++w) { nd += cnt_s[w][0]; ns += cnt_s[w][1]; }
  float t_rgb = 0, t_depth = 0, t_feat = 0, t_sem = 0;
  for (int ray = blockIdx.x * LOSS_WAVES + wave; ray < a.N; ray += gridDim.x * LOSS_WAVES) {
    const float q_im = cur.im, q_gt = cur.gt, q_gd = cur.gd, q_dp = cur.dp, q_sem = cur.sem, q_f = cur.f, q_gf = cur.gf;
    const int label = cur.label;
    cur = fetch(ray + gridDim.x * LOSS_WAVES);   // (the next ray of this wave, if the batch has more rays than the launch has waves)
    if (lane < 3) {
      float diff = q_im - q_gt;
      t_rgb += diff * diff;
      a.g_image[3 * (size_t)ray + lane] = scale * a.w_rgb * 2.f * diff / (3.f * a.N);
    }
    if (lane == 0) {
      float g = 0.f, gd = q_gd;
      if (gd > DEPTH_EPSILON) {
        float diff = q_dp - gd;
        t_depth += fabsf(diff);
        g = scale * a.w_depth * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / (float)nd;
      }
      a.g_depth[ray] = g;
    }
    if (a.g_feat) {
      for (int d = lane; d < a.D; d += 64) {
        float g = 0.f;
        if (a.gt_feat && d < a.Cf) {
          const float fv = d < 64 ? q_f : a.features[(size_t)ray * a.D + d], gv = d < 64 ? q_gf : a.gt_feat[(size_t)ray * a.Cf + d];   // (the first 64 columns came with the batch above)
          float diff = fv - gv;
          t_feat += fabsf(diff);
          g = scale * a.w_feat * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) / ((float)a.N * a.Cf);
        }
        a.g_feat[(size_t)ray * a.D + d] = g;
      }
    }
    if (a.g_sem) {
      if (label >= 0) {
        auto sem_at = [&](int c) { return c < 64 ? q_sem : a.semantic[(size_t)ray * a.C + c]; };   // (c = lane + 64 k: the first chunk is in a register)
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, sem_at(c));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f;
        for (int c = lane; c < a.C; c += 64) se += expf(sem_at(c) - mx);
        se = wave_sum(se);
        for (int c = lane; c < a.C; c += 64) {
          float l = sem_at(c);
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
  if (threadIdx.x == 0) {
    float* slot = a.terms + 8 + 4 * blockIdx.x;
    for (int k = 0; k < 4; ++k) {   // (fixed order: waves 0 .. 15)
      float t = 0.f;
      for (int w = 0; w < LOSS_WAVES; ++w) t += part[w][k];
      slot[k] = t;
    }
    // publish the row, then take a ticket (release fence; the explicit wait keeps the write-back ahead of the ticket)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(a.counts + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = ticket == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (last_s) {   // the whole block folds the rows: thread t takes row t, then a fixed tree (lanes, then waves): same sum every run
    // every reading thread takes the agent-scope acquire itself (a workgroup barrier does not extend one thread's acquire to the
    // others under the HIP memory model, even though one buffer_inv happens to cover the CU's L1 on gfx950); last block only
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < gridDim.x) v = *(const float4*)(a.terms + 8 + 4 * threadIdx.x);   // (gridDim.x <= LOSS_MAX_BLOCKS = 256: waves 0 .. 3 hold rows)
    v.x = wave_sum(v.x); v.y = wave_sum(v.y); v.z = wave_sum(v.z); v.w = wave_sum(v.w);
    __syncthreads();     // (everybody has read `part` above)
    if (lane == 0 && wave < 4) { part[wave][0] = v.x; part[wave][1] = v.y; part[wave][2] = v.z; part[wave][3] = v.w; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float sum[4];
      for (int k = 0; k < 4; ++k) sum[k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
      float r = sum[0] / (3.f * a.N), d = nd ? sum[1] / nd : 0.f, f = (a.gt_feat && a.Cf) ? sum[2] / ((float)a.N * a.Cf) : 0.f,
            sv = ns ? sum[3] / ns : 0.f;
      a.terms[0] = r; a.terms[1] = d; a.terms[2] = f; a.terms[3] = sv;
      a.terms[4] = a.w_rgb * r + a.w_depth * d + a.w_feat * f + a.w_sem * sv;
      a.counts[0] = nd; a.counts[1] = ns;
      __hip_atomic_store(a.counts + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch starts from zero again
    }
  }
}

// terms: 8 + 4 * 256 floats (terms[0..5): rgb, depth, feature, semantic, total; the rest is scratch for the block partial sums);
// counts: 4 int32 ([0] = rays with valid depth, [1] = labelled rays, [2] = arrival ticket: zero before the FIRST launch only)
extern "C" int32_t aln_loss_terms_floats(void) { return 8 + 4 * LOSS_MAX_BLOCKS; }
extern "C" int aln_loss_fwd_bwd(const float* image, const float* depth, const float* semantic, const float* features,
                                const float* gt_rgb, const float* gt_depth, const int32_t* gt_sem, const float* gt_feat,
                                int32_t N, int32_t C, int32_t D, int32_t Cf, float w_rgb, float w_depth, float w_sem,
                                float w_feat, const float* loss_scale, int32_t* counts, float* g_image, float* g_depth,
                                float* g_sem, float* g_feat, float* terms, void* stream) {
  ALN_REQUIRE(image && depth && gt_rgb && gt_depth && gt_sem && counts && g_image && g_depth && terms, "loss: NULL pointer");
  ALN_REQUIRE(!g_sem || semantic, "loss: semantic output missing");
  ALN_REQUIRE(!gt_feat || (features && g_feat && Cf <= D), "loss: feature buffers missing or Cf > D");
  if (N <= 0) return 0;
  LossArgs a{image, depth, semantic, features, gt_rgb, gt_depth, gt_sem, gt_feat, N, C, D, Cf, w_rgb, w_depth, w_sem, w_feat,
             counts, loss_scale, g_image, g_depth, g_sem, g_feat, terms};
  int nb = (N + LOSS_WAVES - 1) / LOSS_WAVES;
  hipLaunchKernelGGL(k_loss, dim3(nb < LOSS_MAX_BLOCKS ? nb : LOSS_MAX_BLOCKS), dim3(64 * LOSS_WAVES), 0, (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("loss");
  return 0;
}
