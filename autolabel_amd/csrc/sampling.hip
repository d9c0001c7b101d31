// Ray sampling and volumetric compositing, forward + backward.
// Replaces the PyTorch op chain of torch-ngp NeRFRenderer.run (fork) reached from
// autolabel/trainer.py:64-70,102-107,127-133, scripts/export.py:83-89, scripts/render.py:96-102:
// near_far_from_aabb, stratified coarse z, sample_pdf (cumsum/searchsorted), sort+gather merge,
// alpha / cumprod weights, weighted sums.  Spec: oracle/nerf_oracle.py (OracleModel.run, sample_pdf).
//
// One wavefront (64-thread block) per ray: the per-ray scans (cumprod, cumsum, suffix sums) are
// chunked scans -- each lane walks a contiguous chunk, lane totals are combined with wave shuffles;
// the coarse/fine merge is a merge-path rank (binary search in LDS) instead of a generic sort.
#include "common.h"
#include <math.h>

#define MAX_S 1024  // max samples per ray per pass

// ------------------------------------------------------------------ near/far vs the [-bound, bound]^3 box
// (torch-ngp raymarching.near_far_from_aabb; a miss gives near = far = min_near) -- oracle: OracleModel.near_far
__device__ inline void ray_aabb(const float* o3, const float* d3, float bound, float min_near, float& near, float& far) {
  float tn = -INFINITY, tf = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float o = o3[k], d = d3[k];
    float inv = __fdiv_rn(1.0f, d);
    float t1 = __fmul_rn(__fsub_rn(-bound, o), inv), t2 = __fmul_rn(__fsub_rn(bound, o), inv);
    tn = fmaxf(tn, fminf(t1, t2));
    tf = fminf(tf, fmaxf(t1, t2));
  }
  bool miss = !(tn <= tf);
  near = miss ? min_near : fmaxf(tn, min_near);
  far = fmaxf(miss ? min_near : tf, near);
}

__global__ void k_ray_aabb(const float* __restrict__ ro, const float* __restrict__ rd, int N, float bound, float min_near,
                           float* __restrict__ nears, float* __restrict__ fars) {
  for (int ray = blockIdx.x * blockDim.x + threadIdx.x; ray < N; ray += gridDim.x * blockDim.x) {
    float near, far;
    ray_aabb(ro + 3 * (size_t)ray, rd + 3 * (size_t)ray, bound, min_near, near, far);
    nears[ray] = near; fars[ray] = far;
  }
}

extern "C" int aln_ray_aabb(const float* rays_o, const float* rays_d, int32_t N, float bound, float min_near, float* nears,
                            float* fars, void* stream) {
  ALN_REQUIRE(rays_o && rays_d && nears && fars, "ray_aabb: NULL pointer");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_ray_aabb, dim3(aln_grid_for(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, N, bound,
                     min_near, nears, fars);
  ALN_CHECK_LAUNCH("ray_aabb");
  return 0;
}

// ------------------------------------------------------------------ coarse
__global__ void k_sample_coarse(const float* __restrict__ ro, const float* __restrict__ rd, int N, int S1, float bound,
                                float min_near, int perturb, uint32_t seed, uint32_t step, const uint32_t* __restrict__ step_dev,
                                const float* __restrict__ noise, float* __restrict__ nears, float* __restrict__ fars,
                                float* __restrict__ z) {
  // step_dev: the step number lives in device memory (added to `step`) so that a captured hipGraph replays with fresh noise
  const uint32_t key = aln_rand_key(seed, ALN_STREAM_PERTURB, step + (step_dev ? *step_dev : 0u));
  size_t total = (size_t)N * S1;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    int ray = (int)(t / S1), i = (int)(t % S1);
    float near, far;
    ray_aabb(ro + 3 * (size_t)ray, rd + 3 * (size_t)ray, bound, min_near, near, far);
    if (i == 0) { nears[ray] = near; fars[ray] = far; }
    float lin = __fdiv_rn((float)i, (float)(S1 > 1 ? S1 - 1 : 1));
    float span = __fsub_rn(far, near);
    float zz = __fadd_rn(near, __fmul_rn(span, lin));
    if (perturb) {
      float u = noise ? noise[t] : aln_rand_uniform(key, (uint32_t)t);
      float sd = __fdiv_rn(span, (float)S1);
      zz = __fadd_rn(zz, __fmul_rn(__fsub_rn(u, 0.5f), sd));
    }
    z[t] = zz;
  }
}

extern "C" int aln_sample_coarse(const float* rays_o, const float* rays_d, int32_t N, int32_t S1, float bound,
                                 float min_near, int32_t perturb, uint32_t seed, uint32_t step, const float* noise,
                                 float* nears, float* fars, float* z, const uint32_t* step_dev, void* stream) {
  ALN_REQUIRE(rays_o && rays_d && nears && fars && z && S1 > 0 && S1 <= MAX_S, "sample_coarse: bad arguments");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_sample_coarse, dim3(aln_grid_for((int64_t)N * S1, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, N, S1, bound, min_near, perturb, seed, step, step_dev, noise, nears, fars, z);
  ALN_CHECK_LAUNCH("sample_coarse");
  return 0;
}

// ------------------------------------------------------------------ wave-level chunked scans over LDS arrays
// T[i] = prod_{j<i} v[j]   (exclusive), one wave
__device__ inline void scan_excl_prod(const float* v, float* T, int n, int lane) {
  int ipl = (n + 63) / 64, s = lane * ipl, e = min(s + ipl, n);
  float loc = 1.f;
  for (int i = s; i < e; ++i) loc *= v[i];
  float inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(inc, o); if (lane >= o) inc *= t; }
  float run = __shfl_up(inc, 1);
  if (lane == 0) run = 1.f;
  for (int i = s; i < e; ++i) { T[i] = run; run *= v[i]; }
}
// C[i] = sum_{j<=i} v[j]  (inclusive)
__device__ inline void scan_incl_sum(const float* v, float* C, int n, int lane) {
  int ipl = (n + 63) / 64, s = lane * ipl, e = min(s + ipl, n);
  float loc = 0.f;
  for (int i = s; i < e; ++i) loc += v[i];
  float inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  float run = __shfl_up(inc, 1);
  if (lane == 0) run = 0.f;
  for (int i = s; i < e; ++i) { run += v[i]; C[i] = run; }
}
// R[i] = sum_{j>i} v[j]   (exclusive suffix)
__device__ inline void scan_suffix_excl(const float* v, float* R, int n, int lane) {
  int ipl = (n + 63) / 64, s = lane * ipl, e = min(s + ipl, n);
  float loc = 0.f;
  for (int i = s; i < e; ++i) loc += v[i];
  float inc = loc;  // inclusive suffix over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { float t = __shfl_down(inc, o); if (lane + o < 64) inc += t; }
  float run = __shfl_down(inc, 1);
  if (lane == 63) run = 0.f;
  for (int i = e - 1; i >= s; --i) { R[i] = run; run += v[i]; }
}

// ------------------------------------------------------------------ fine (importance) samples
__global__ __launch_bounds__(64) void k_sample_fine(const float* __restrict__ zc, const float* __restrict__ sigma,
                                                   const float* __restrict__ nears, const float* __restrict__ fars, int N,
                                                   int S1, int S2, float density_scale, int perturb, uint32_t seed, uint32_t step, const uint32_t* __restrict__ step_dev,
                                                   const float* __restrict__ u_in, float* __restrict__ zf) {
  const uint32_t key = aln_rand_key(seed, ALN_STREAM_PDF, step + (step_dev ? *step_dev : 0u));
  extern __shared__ float sm[];
  float* z = sm;            // [S1]
  float* a = z + S1;        // [S1]  val / pdf scratch
  float* b = a + S1;        // [S1]  T / cdf
  float* w = b + S1;        // [S1]
  int S2p = 1; while (S2p < S2) S2p <<= 1;
  float* u = w + S1;        // [S2p]
  const int lane = threadIdx.x;
  for (int ray = blockIdx.x; ray < N; ray += gridDim.x) {
    // (round 6: the ray's distances and densities are requested together, before near / far are used -- one HBM round trip at the top of
    //  the wave's chain instead of one per 64 samples and array; see k_composite_fwd)
    constexpr int KU = 4;
    const bool fast = S1 <= 64 * KU;
    float zr[KU], sr[KU];
    if (fast) {
#pragma unroll
      for (int q = 0; q < KU; ++q) { const int i = min(lane + 64 * q, S1 - 1); zr[q] = zc[(size_t)ray * S1 + i]; sr[q] = sigma[(size_t)ray * S1 + i]; }
    }
    const float sd = __fdiv_rn(__fsub_rn(fars[ray], nears[ray]), (float)S1);
    if (fast) {
#pragma unroll
      for (int q = 0; q < KU; ++q) { const int i = lane + 64 * q; if (i < S1) z[i] = zr[q]; }
    } else {
      for (int i = lane; i < S1; i += 64) z[i] = zc[(size_t)ray * S1 + i];
    }
    __syncthreads();
    auto alpha_of = [&](int i, float sigma_i) {
      float delta = (i + 1 < S1) ? z[i + 1] - z[i] : sd;
      float alpha = 1.f - expf(-delta * density_scale * sigma_i);
      w[i] = alpha;
      a[i] = 1.f - alpha + 1e-15f;
    };
    if (fast) {
#pragma unroll
      for (int q = 0; q < KU; ++q) { const int i = lane + 64 * q; if (i < S1) alpha_of(i, sr[q]); }
    } else {
      for (int i = lane; i < S1; i += 64) alpha_of(i, sigma[(size_t)ray * S1 + i]);
    }
    __syncthreads();
    scan_excl_prod(a, b, S1, lane);
    __syncthreads();
    for (int i = lane; i < S1; i += 64) w[i] = w[i] * b[i];
    __syncthreads();
    // pdf over weights[1:-1] (+1e-5), cdf with leading 0 -> length S1-1
    const int nb = S1 - 2;
    float part = 0.f;
    for (int k = lane; k < nb; k += 64) { a[k] = w[k + 1] + 1e-5f; part += a[k]; }
    float tot = wave_sum(part);
    __syncthreads();
    for (int k = lane; k < nb; k += 64) a[k] = a[k] / tot;
    __syncthreads();
    scan_incl_sum(a, b + 1, nb, lane);
    if (lane == 0) b[0] = 0.f;
    // uniforms
    for (int j = lane; j < S2p; j += 64) {
      float v = INFINITY;
      if (j < S2) {
        if (!perturb) v = __fdiv_rn((float)j + 0.5f, (float)S2);
        else v = u_in ? u_in[(size_t)ray * S2 + j] : aln_rand_uniform(key, (uint32_t)((size_t)ray * S2 + j));
      }
      u[j] = v;
    }
    __syncthreads();
    if (perturb) {  // bitonic sort ascending
      for (int k = 2; k <= S2p; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = lane; i < S2p; i += 64) {
            int ixj = i ^ j;
            if (ixj > i) {
              float x = u[i], y = u[ixj];
              bool up = (i & k) == 0;
              if ((x > y) == up) { u[i] = y; u[ixj] = x; }
            }
          }
          __syncthreads();
        }
    }
    const int nc = S1 - 1;  // cdf / bins length
    for (int j = lane; j < S2; j += 64) {
      float uu = u[j];
      int lo = 0, hi = nc;  // first index with cdf > u  (searchsorted right=True)
      while (lo < hi) { int mid = (lo + hi) >> 1; if (b[mid] > uu) hi = mid; else lo = mid + 1; }
      int below = max(lo - 1, 0), above = min(lo, nc - 1);
      float cb = b[below], ca = b[above];
      float bb = z[below] + 0.5f * (z[below + 1] - z[below]);
      float ba = z[above] + 0.5f * (z[above + 1] - z[above]);
      float den = ca - cb;
      if (den < 1e-5f) den = 1.f;
      float t = (uu - cb) / den;
      zf[(size_t)ray * S2 + j] = bb + t * (ba - bb);
    }
    __syncthreads();
  }
}

extern "C" int aln_sample_fine(const float* z_coarse, const float* sigma_coarse, const float* nears, const float* fars,
                               int32_t N, int32_t S1, int32_t S2, float density_scale, int32_t perturb, uint32_t seed,
                               uint32_t step, const float* u, float* z_fine, const uint32_t* step_dev, void* stream) {
  ALN_REQUIRE(z_coarse && sigma_coarse && nears && fars && z_fine, "sample_fine: NULL pointer");
  ALN_REQUIRE(S1 >= 3 && S1 <= MAX_S && S2 > 0 && S2 <= MAX_S, "sample_fine: S1=%d S2=%d out of range", S1, S2);
  if (N <= 0) return 0;
  int S2p = 1; while (S2p < S2) S2p <<= 1;
  size_t lds = (size_t)(4 * S1 + S2p) * sizeof(float);
  hipLaunchKernelGGL(k_sample_fine, dim3(N < 65535 ? N : 65535), dim3(64), lds, (hipStream_t)stream, z_coarse, sigma_coarse,
                     nears, fars, N, S1, S2, density_scale, perturb, seed, step, step_dev, u, z_fine);
  ALN_CHECK_LAUNCH("sample_fine");
  return 0;
}

// ------------------------------------------------------------------ compositing
// local sample id: < S1 coarse i, else fine j = id - S1.  row(ray, id) in pass-major order.
__device__ inline size_t row_of(int ray, int id, int N, int S1, int S2) {
  return id < S1 ? (size_t)ray * S1 + id : (size_t)N * S1 + (size_t)ray * S2 + (id - S1);
}

struct CompFwd {
  const float* ro; const float* rd; const float* norms; const float* nears; const float* fars;
  const float* z;      // [M] rows
  const float* sigma;  // [M] rows
  int N, S1, S2; float bound, density_scale;
  uint16_t* perm;      // [N, S]
  float* w_row; float* T_row; float* delta_row;  // [M]
  float* wsum; float* depth; float* depth_var; float* coords;  // [N], [N], [N], [N,3]
  const float* delta_in;   // [N, S1] explicit step lengths (occupancy-grid marching, S2 == 0) or NULL: differences of z
};

__global__ __launch_bounds__(64) void k_composite_fwd(CompFwd p) {
  extern __shared__ float sm[];
  const int S = p.S1 + p.S2, lane = threadIdx.x;
  float* zs = sm;          // sorted z [S]
  float* sg = zs + S;      // sorted sigma
  float* v = sg + S;       // 1-alpha+eps
  float* T = v + S;
  float* zin = T + S;      // unsorted z [S]
  uint16_t* lid = (uint16_t*)(zin + S);  // [S]
  constexpr int KU = 4;   // samples per lane whose inputs are requested up front (the training step: 256 samples per ray)
  const bool fast = S <= 64 * KU;
  for (int ray = blockIdx.x; ray < p.N; ray += gridDim.x) {
    // One wave walks one ray and every ray of a batch is resident at once: the kernel's time is the length of this wave's chain of HBM
    // round trips.  Round 6: everything the ray reads -- distances, densities, near / far, the norm, origin and direction -- is requested
    // HERE, before the first value is used (it was: four serial trips for z, four for sigma behind the rank searches, three for the scalars).
    float zr[KU], sr[KU];
    if (fast) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int i = lane + 64 * u;
        const size_t row = row_of(ray, i < S ? i : 0, p.N, p.S1, p.S2);
        zr[u] = p.z[row]; sr[u] = p.sigma[row];
      }
    }
    const float near_ = p.nears[ray], far_ = p.fars[ray], norm_ = p.norms[ray];
    float ro[3], rd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ro[k] = p.ro[3 * (size_t)ray + k]; rd[k] = p.rd[3 * (size_t)ray + k]; }
    // (a NaN distance -- a diverged field -- would make the two rank rules below inconsistent and leave slots of the
    //  permutation unwritten; it is ordered like +inf instead.  Finite inputs are untouched.)
    if (fast) {
#pragma unroll
      for (int u = 0; u < KU; ++u) { const int i = lane + 64 * u; if (i < S) zin[i] = zr[u] == zr[u] ? zr[u] : 3.0e38f; }
    } else {
      for (int i = lane; i < S; i += 64) { const float zz = p.z[row_of(ray, i, p.N, p.S1, p.S2)]; zin[i] = zz == zz ? zz : 3.0e38f; }
    }
    __syncthreads();
    const float* zc = zin; const float* zf = zin + p.S1;
    auto place = [&](int i, float sigma_i) {
      int pos;
      if (i < p.S1) {  // # fine strictly less
        float x = zc[i]; int lo = 0, hi = p.S2;
        while (lo < hi) { int m = (lo + hi) >> 1; if (zf[m] < x) lo = m + 1; else hi = m; }
        pos = i + lo;
      } else {  // # coarse <=
        float x = zf[i - p.S1]; int lo = 0, hi = p.S1;
        while (lo < hi) { int m = (lo + hi) >> 1; if (zc[m] <= x) lo = m + 1; else hi = m; }
        pos = (i - p.S1) + lo;
      }
      zs[pos] = zin[i];
      sg[pos] = sigma_i;
      lid[pos] = (uint16_t)i;
    };
    if (fast) {
#pragma unroll
      for (int u = 0; u < KU; ++u) { const int i = lane + 64 * u; if (i < S) place(i, sr[u]); }
    } else {
      for (int i = lane; i < S; i += 64) place(i, p.sigma[row_of(ray, i, p.N, p.S1, p.S2)]);
    }
    __syncthreads();
    const float sd = __fdiv_rn(__fsub_rn(far_, near_), (float)p.S1);
    for (int k = lane; k < S; k += 64) {
      float delta = p.delta_in ? p.delta_in[(size_t)ray * S + k] : ((k + 1 < S) ? zs[k + 1] - zs[k] : sd);
      float alpha = 1.f - expf(-delta * p.density_scale * sg[k]);
      v[k] = 1.f - alpha + 1e-15f;
      sg[k] = alpha;       // reuse: alpha
      zin[k] = delta;      // reuse: delta
    }
    __syncthreads();
    scan_excl_prod(v, T, S, lane);
    __syncthreads();
    const float inv_norm = 1.0f / norm_;
    float a_w = 0.f, a_d = 0.f, a_c[3] = {0, 0, 0};
    for (int k = lane; k < S; k += 64) {
      float w = sg[k] * T[k];
      size_t row = row_of(ray, lid[k], p.N, p.S1, p.S2);
      p.w_row[row] = w; p.T_row[row] = T[k]; p.delta_row[row] = zin[k];
      p.perm[(size_t)ray * S + k] = lid[k];
      float x[3];
      aln_sample_xyz(ro, rd, zs[k], p.bound, x);
      a_w += w; a_d += w * (zs[k] * inv_norm);
      a_c[0] += w * x[0]; a_c[1] += w * x[1]; a_c[2] += w * x[2];
      v[k] = w;  // keep for variance pass
    }
    a_w = wave_sum(a_w); a_d = wave_sum(a_d);
    a_c[0] = wave_sum(a_c[0]); a_c[1] = wave_sum(a_c[1]); a_c[2] = wave_sum(a_c[2]);
    float var = 0.f;
    for (int k = lane; k < S; k += 64) { float e = zs[k] * inv_norm - a_d; var += v[k] * e * e; }
    var = wave_sum(var);
    if (lane == 0) {
      p.wsum[ray] = a_w; p.depth[ray] = a_d; p.depth_var[ray] = var;
      p.coords[3 * (size_t)ray] = a_c[0]; p.coords[3 * (size_t)ray + 1] = a_c[1]; p.coords[3 * (size_t)ray + 2] = a_c[2];
    }
    __syncthreads();
  }
}

extern "C" int aln_composite_fwd(const float* rays_o, const float* rays_d, const float* norms, const float* nears,
                                 const float* fars, const float* z, const float* sigma, int32_t N, int32_t S1, int32_t S2,
                                 float bound, float density_scale, uint16_t* perm, float* w_row, float* T_row,
                                 float* delta_row, float* wsum, float* depth, float* depth_var, float* coords,
                                 const float* delta_in, void* stream) {
  ALN_REQUIRE(rays_o && rays_d && norms && nears && fars && z && sigma && perm && w_row && T_row && delta_row && wsum &&
                  depth && depth_var && coords, "composite_fwd: NULL pointer");
  ALN_REQUIRE(S1 > 0 && S2 >= 0 && S1 + S2 <= 2 * MAX_S, "composite_fwd: sample counts out of range");
  if (N <= 0) return 0;
  CompFwd p{rays_o, rays_d, norms, nears, fars, z, sigma, N, S1, S2, bound, density_scale,
            perm, w_row, T_row, delta_row, wsum, depth, depth_var, coords, delta_in};
  ALN_REQUIRE(!delta_in || S2 == 0, "composite_fwd: explicit step lengths need a single pass (S2 == 0)");
  int S = S1 + S2;
  size_t lds = (size_t)S * (5 * sizeof(float) + sizeof(uint16_t)) + 16;
  hipLaunchKernelGGL(k_composite_fwd, dim3(N < 65535 ? N : 65535), dim3(64), lds, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("composite_fwd");
  return 0;
}

// image / semantic / features sums.  color_out is the compacted color-MLP output (pre-sigmoid, fp16,
// [n_live,16]); cidx_row maps row -> compact index (-1 = masked, contributes 0: models.py:195-203).
struct CompOut {
  const float* w_row; const int* cidx_row; const h16* color_out; const h16* logits; const h16* feat;
  const float* wsum;
  int N, S1, S2, C, Cpad, D; float bg;
  float* image; float* semantic; float* features;
  const float* tile_sums;   // optional [M / 32][96]: per-tile sums of w * f (64) and w * logits (<= 32) left by aln_sem_heads_fwd_sums
  const float* feat_sums;   // optional [M / 32][D]: per-tile sums of w * feat left by the producer of feat (aln_wide_nt_gen): feat rows are not read
};

__global__ __launch_bounds__(64) void k_composite_out(CompOut p) {
  const int S = p.S1 + p.S2, lane = threadIdx.x;
  for (int ray = blockIdx.x; ray < p.N; ray += gridDim.x) {
    // rgb: lanes split the samples, wave-reduce.  Four samples per lane at a time, every load UNCONDITIONAL (clamped sample / compact
    // index, the value dropped afterwards): the whole kernel is one wave's chain of HBM round trips -- all rays are resident at once -- and
    // with a load per branch hipcc waits for each in turn: eight dependent round trips for 256 samples, now two (round 6).
    float c0 = 0, c1 = 0, c2 = 0;
    for (int k0 = 0; k0 < S; k0 += 256) {
      int ci[4]; float w[4]; h16x4 o[4]; bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 64 * u + lane;
        ok[u] = k < S;
        const size_t row = row_of(ray, ok[u] ? k : 0, p.N, p.S1, p.S2);
        ci[u] = p.cidx_row[row]; w[u] = p.w_row[row];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = *(const h16x4*)(p.color_out + (size_t)max(ci[u], 0) * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u] && ci[u] >= 0) {
          c0 += w[u] / (1.f + expf(-(float)o[u][0])); c1 += w[u] / (1.f + expf(-(float)o[u][1])); c2 += w[u] / (1.f + expf(-(float)o[u][2]));
        }
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    if (lane == 0) {
      float r = (1.f - p.wsum[ray]) * p.bg;
      p.image[3 * (size_t)ray] = c0 + r; p.image[3 * (size_t)ray + 1] = c1 + r; p.image[3 * (size_t)ray + 2] = c2 + r;
    }
    if (p.tile_sums) {   // the ray's tiles in a fixed order: coarse pass, then fine pass (bit-reproducible)
      const int t1 = p.S1 / 32, t2 = p.S2 / 32;
      const float* a1 = p.tile_sums + (size_t)ray * t1 * 96;
      const float* a2 = p.tile_sums + ((size_t)p.N * t1 + (size_t)ray * t2) * 96;
      for (int ch = lane; ch < 96; ch += 64) {
        float a = 0.f;
        // (four tiles requested at a time, added in tile order: the same sum, a quarter of the round trips)
        auto tiles = [&](const float* q, int nt) {
          int t = 0;
          for (; t + 4 <= nt; t += 4) {
            const float v0 = q[t * 96 + ch], v1 = q[(t + 1) * 96 + ch], v2 = q[(t + 2) * 96 + ch], v3 = q[(t + 3) * 96 + ch];
            a += v0; a += v1; a += v2; a += v3;
          }
          for (; t < nt; ++t) a += q[t * 96 + ch];
        };
        tiles(a1, t1); tiles(a2, t2);
        if (ch < 64) { if (p.features) p.features[(size_t)ray * p.D + ch] = a; }
        else if (ch - 64 < p.C && p.semantic) p.semantic[(size_t)ray * p.C + ch - 64] = a;
      }
    }
    if (p.feat_sums) {   // the ray's tiles in a fixed order: coarse pass, then fine pass
      const int t1 = p.S1 / 32, t2 = p.S2 / 32;
      const float* a1 = p.feat_sums + (size_t)ray * t1 * p.D;
      const float* a2 = p.feat_sums + ((size_t)p.N * t1 + (size_t)ray * t2) * p.D;
      for (int ch = lane; ch < p.D; ch += 64) {
        float a = 0.f;
        for (int t = 0; t < t1; ++t) a += a1[(size_t)t * p.D + ch];
        for (int t = 0; t < t2; ++t) a += a2[(size_t)t * p.D + ch];
        p.features[(size_t)ray * p.D + ch] = a;
      }
    }
    // channel sums: `lpr` lanes share a row (one 16-byte chunk each), so one wave instruction reads 64 / lpr whole rows;
    // every lane keeps 8 channel sums over its rows and the row groups are folded with cross-lane adds at the end
    if (p.logits || p.feat) {
      const int nch_f = p.feat ? p.D / 8 : 0, nch_l = p.logits ? p.Cpad / 8 : 0;
      int lpr = 1;
      while (lpr < 64 && lpr < max(nch_f, nch_l)) lpr <<= 1;
      const int cg = lane & (lpr - 1), rg = lane / lpr, rpi = 64 / lpr;
      for (int pass = 0; pass < 2; ++pass) {
        const h16* src = pass ? p.feat : p.logits;
        if (!src) continue;    // (feature-only compositing: the linear LSeg path sums hidden activations, no logits)
        const int ld = pass ? p.D : p.Cpad, nch = pass ? nch_f : nch_l, nout = pass ? p.D : p.C;
        float* dst = (pass ? p.features : p.semantic) + (size_t)ray * nout;
        for (int ch = cg; ch < nch; ch += lpr) {
          float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
          for (int k = rg; k < S; k += rpi) {
            const size_t row = row_of(ray, k, p.N, p.S1, p.S2);
            const float w = p.w_row[row];
            const h16x8 v = *(const h16x8*)(src + row * ld + 8 * ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * (float)v[j];
          }
          for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o);
          }
          if (rg == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (8 * ch + j < nout) dst[8 * ch + j] = acc[j];
          }
        }
      }
    }
  }
}

extern "C" int aln_composite_out(const float* w_row, const int32_t* cidx_row, const void* color_out, const void* logits,
                                 const void* feat, const float* wsum, int32_t N, int32_t S1, int32_t S2, int32_t C,
                                 int32_t Cpad, int32_t D, float bg, float* image, float* semantic, float* features,
                                 const float* tile_sums, void* stream) {
  ALN_REQUIRE(w_row && cidx_row && color_out && wsum && image, "composite_out: NULL pointer");
  ALN_REQUIRE(!tile_sums || (!logits && !feat && D == 64 && C <= 32 && S1 % 32 == 0 && S2 % 32 == 0 && semantic && features),
              "composite_out: tile sums replace the logits / feature rows (D = 64, <= 32 classes, sample counts multiples of 32)");
  ALN_REQUIRE((!logits || semantic) && (!feat || features), "composite_out: semantic / feature output buffers missing");
  ALN_REQUIRE((!logits || Cpad % 8 == 0) && (!feat || D % 8 == 0), "composite_out: Cpad and D must be multiples of 8");
  if (N <= 0) return 0;
  CompOut p{w_row, cidx_row, (const h16*)color_out, (const h16*)logits, (const h16*)feat, wsum, N, S1, S2, C, Cpad, D, bg,
            image, semantic, features, tile_sums, nullptr};
  hipLaunchKernelGGL(k_composite_out, dim3(N < 65535 ? N : 65535), dim3(64), 0, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("composite_out");
  return 0;
}
// image as aln_composite_out; features[ray][0 .. D) = the ray's per-tile sums feat_sums[M / 32][D] added up in a fixed order (the producer
// of the per-sample activation left them: aln_wide_nt_gen).  No per-sample rows are read.  Sample counts must be multiples of 32.
extern "C" int aln_composite_out_featsums(const float* w_row, const int32_t* cidx_row, const void* color_out, const float* wsum, int32_t N,
                                          int32_t S1, int32_t S2, int32_t D, float bg, float* image, float* features, const float* feat_sums,
                                          void* stream) {
  ALN_REQUIRE(w_row && cidx_row && color_out && wsum && image && features && feat_sums, "composite_out_featsums: NULL pointer");
  ALN_REQUIRE(S1 % 32 == 0 && S2 % 32 == 0 && D > 0, "composite_out_featsums: sample counts must be multiples of 32");
  if (N <= 0) return 0;
  CompOut p{w_row, cidx_row, (const h16*)color_out, nullptr, nullptr, wsum, N, S1, S2, 0, 0, D, bg, image, nullptr, features, nullptr, feat_sums};
  hipLaunchKernelGGL(k_composite_out, dim3(N < 65535 ? N : 65535), dim3(64), 0, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("composite_out_featsums");
  return 0;
}

// ------------------------------------------------------------------ backward
struct CompBwd {
  const float* norms; const float* z; const float* sigma;
  const uint16_t* perm; const float* w_row; const float* T_row; const float* delta_row;
  const int* cidx_row; const h16* color_out; const h16* logits; const h16* feat; const h16* sigma_out;  // [M,16], col 0 = h0
  const float* g_image; const float* g_depth; const float* g_sem; const float* g_feat;  // per ray, already loss-scaled
  int N, S1, S2, C, Cpad, D, mask_feat; float bg, density_scale;   // mask_feat: d_feat rows *= (feat > 0) (feat = a post-ReLU activation)
  float* d_h0;         // [M] fp32
  h16* d_color_out;    // [n_live,16]
  h16* d_logits;       // [M,Cpad]
  h16* d_feat;         // [M,D]
  int* found_inf;
  const float* dots_row;   // optional [M]: <logits_s, g_sem[ray]> + <f_s, g_feat[ray]> already computed (aln_sem_heads_bwd): logits / feat are not read
};

#ifndef CB_KU
#define CB_KU 4    // row trips requested together by a wave of k_composite_bwd (parts 2 / 3: one lane per row)
#endif
#ifndef CB_KU1
#define CB_KU1 2   // ... part 1 (lpr lanes per row).  Measured (4096 / 1024 rays): old kernel 67 / 47 us; KU1 4 at 3 waves per SIMD 69 / 29; KU1 2 at 4 waves 53 / 31
#endif
#ifndef CB_WPE
#define CB_WPE 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CB_WPE))) void k_composite_bwd(CompBwd p) {
  extern __shared__ float sm[];
  const int S = p.S1 + p.S2, lane = threadIdx.x;
  float* P = sm;          // sorted dw*w [S]
  float* R = P + S;       // suffix
  float* dws = R + S;     // sorted dw
  float* gs = dws + S;    // g_sem [C]
  float* gf = gs + p.Cpad;  // g_feat [D]
  float* dsem = gf + p.D;   // d(loss)/dw from the semantic outputs, by sample id [S]
  bool bad = false;
  // the <logits, g_sem> + <f, g_feat> dot products (and, for the library-GEMM heads, the d_logits / d_feat rows = w * g)
  // run row-major with `lpr` lanes per row (16-byte chunks, whole rows per instruction) instead of one strided row per lane
  const bool rowmajor_dots = p.logits != nullptr || p.feat != nullptr;
  const int nch_f = p.feat ? p.D / 8 : 0, nch_l = p.logits ? p.Cpad / 8 : 0;
  int lpr = 1;
  while (lpr < 64 && lpr < max(nch_f, nch_l)) lpr <<= 1;
  const int cg = lane & (lpr - 1), rg = lane / lpr, rpi = 64 / lpr;
  for (int ray = blockIdx.x; ray < p.N; ray += gridDim.x) {
    const float gi0 = p.g_image[3 * (size_t)ray], gi1 = p.g_image[3 * (size_t)ray + 1], gi2 = p.g_image[3 * (size_t)ray + 2];
    const float gd = p.g_depth[ray] / p.norms[ray];
    if (rowmajor_dots) {
      for (int c = lane; c < p.Cpad; c += 64) gs[c] = (p.logits && c < p.C) ? p.g_sem[(size_t)ray * p.C + c] : 0.f;
      for (int d = lane; d < p.D; d += 64) gf[d] = (p.feat && p.g_feat) ? p.g_feat[(size_t)ray * p.D + d] : 0.f;
    }
    __syncthreads();
    // One wave walks one ray, so the kernel's time is the length of this wave's dependent chain, not a byte count: with one row
    // group per trip and the loads of a trip issued after the previous trip's shuffles, a ray took ~41 us however few rays there
    // were (profiles/r03_train_kernel_stats_B1024: 47 us at 1024 rays, 67 us at 4096).  All three parts below therefore request the
    // rows of CB_KU trips before touching any of them, and part 3's operands ride along with part 2's (same rows).
    if (rowmajor_dots) {
      if (nch_l <= lpr && nch_f <= lpr) {   // one chunk of each kind per lane (always, up to 512 columns)
        const bool hl = cg < nch_l, hfe = cg < nch_f;
        float gsr[8], gfr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gsr[j] = hl ? gs[8 * cg + j] : 0.f; gfr[j] = hfe ? gf[8 * cg + j] : 0.f; }
        const bool need_w = p.d_logits || p.d_feat;
        for (int k0 = rg; k0 < S; k0 += CB_KU1 * rpi) {
          h16x8 vl[CB_KU1], vf[CB_KU1]; float wv[CB_KU1]; size_t rowv[CB_KU1]; bool ok[CB_KU1];
#pragma unroll
          for (int u = 0; u < CB_KU1; ++u) {
            const int k = k0 + u * rpi;
            ok[u] = k < S;
            rowv[u] = row_of(ray, ok[u] ? k : 0, p.N, p.S1, p.S2);
            const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            // (unconditional loads -- the row is a valid one for every lane, the chunk clamped; see the note at part 2)
            vl[u] = zero; vf[u] = zero;
            if (nch_l) { const h16x8 q = *(const h16x8*)(p.logits + rowv[u] * p.Cpad + 8 * min(cg, nch_l - 1)); vl[u] = (hl && ok[u]) ? q : zero; }
            if (nch_f) { const h16x8 q = *(const h16x8*)(p.feat + rowv[u] * p.D + 8 * min(cg, nch_f - 1)); vf[u] = (hfe && ok[u]) ? q : zero; }
            { const float q = p.w_row[rowv[u]]; wv[u] = (need_w && ok[u]) ? q : 0.f; }
          }
#pragma unroll
          for (int u = 0; u < CB_KU1; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)vl[u][j] * gsr[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)vf[u][j] * gfr[j];
            if (p.d_logits && hl && ok[u]) {   // library-GEMM heads: d(logits) = w * g_sem, d(f) = w * g_feat as whole rows
              h16x8 o8;
#pragma unroll
              for (int j = 0; j < 8; ++j) { o8[j] = (h16)(wv[u] * gsr[j]); bad |= !(fabsf((float)o8[j]) <= 65504.f); }
              *(h16x8*)(p.d_logits + rowv[u] * p.Cpad + 8 * cg) = o8;
            }
            if (p.d_feat && hfe && ok[u]) {
              h16x8 o8;
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                o8[j] = (p.mask_feat && !((float)vf[u][j] > 0.f)) ? (h16)0.f : (h16)(wv[u] * gfr[j]);
                bad |= !(fabsf((float)o8[j]) <= 65504.f);
              }
              *(h16x8*)(p.d_feat + rowv[u] * p.D + 8 * cg) = o8;
            }
            for (int o = 1; o < lpr; o <<= 1) acc += __shfl_xor(acc, o);
            if (cg == 0 && ok[u]) dsem[k0 + u * rpi] = acc;
          }
        }
      } else {
        for (int k = rg; k < S; k += rpi) {   // S % rpi == 0 is not required: k only feeds loads and the final store
          const size_t row = row_of(ray, k, p.N, p.S1, p.S2);
          float acc = 0.f;
          for (int ch = cg; ch < nch_l; ch += lpr) {
            const h16x8 v = *(const h16x8*)(p.logits + row * p.Cpad + 8 * ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[j] * gs[8 * ch + j];
          }
          for (int ch = cg; ch < nch_f; ch += lpr) {
            const h16x8 v = *(const h16x8*)(p.feat + row * p.D + 8 * ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[j] * gf[8 * ch + j];
          }
          if (p.d_logits || p.d_feat) {
            const float w = p.w_row[row];
            for (int ch = cg; ch < nch_l && p.d_logits; ch += lpr) {
              h16x8 o8;
#pragma unroll
              for (int j = 0; j < 8; ++j) { o8[j] = (h16)(w * gs[8 * ch + j]); bad |= !(fabsf((float)o8[j]) <= 65504.f); }
              *(h16x8*)(p.d_logits + row * p.Cpad + 8 * ch) = o8;
            }
            for (int ch = cg; ch < nch_f && p.d_feat; ch += lpr) {
              h16x8 o8, fv;
              if (p.mask_feat) fv = *(const h16x8*)(p.feat + row * p.D + 8 * ch);
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                o8[j] = (p.mask_feat && !((float)fv[j] > 0.f)) ? (h16)0.f : (h16)(w * gf[8 * ch + j]);
                bad |= !(fabsf((float)o8[j]) <= 65504.f);
              }
              *(h16x8*)(p.d_feat + row * p.D + 8 * ch) = o8;
            }
          }
          for (int o = 1; o < lpr; o <<= 1) acc += __shfl_xor(acc, o);
          if (cg == 0) dsem[k] = acc;
        }
      }
      __syncthreads();
    }
    const int nit = (S + 63) >> 6;
    const bool keep = nit <= CB_KU;    // part 3 visits the rows of part 2: its operands are requested here when they fit the registers
    size_t krow[CB_KU]; float kdel[CB_KU], kT[CB_KU], ksig[CB_KU]; h16 kh0h[CB_KU];
    for (int i0 = 0; i0 < nit; i0 += CB_KU) {
      int id[CB_KU]; size_t rowv[CB_KU]; bool ok[CB_KU]; float wv[CB_KU], zv[CB_KU], dv[CB_KU]; int ci[CB_KU]; h16x4 ov[CB_KU];
#pragma unroll
      for (int u = 0; u < CB_KU; ++u) {
        const int k = lane + 64 * (i0 + u);
        ok[u] = k < S;
        const int q = (int)p.perm[(size_t)ray * S + (ok[u] ? k : 0)];
        id[u] = ok[u] ? q : 0;
      }
      __builtin_amdgcn_sched_barrier(0);   // (all CB_KU sample ids are requested before the first is waited for: the scheduler would sink each to its use)
      // Every load below is UNCONDITIONAL (round 6): lanes past the ray's samples read row `id 0` of the ray -- a valid row -- and drop the
      // value.  With `ok ? load : 0` hipcc put each load in a branch of its own and, no longer knowing how many younger loads were in
      // flight, waited for every one of them in turn (s_waitcnt vmcnt(0)): the batching of CB_KU trips had never overlapped anything.
#pragma unroll
      for (int u = 0; u < CB_KU; ++u) {
        rowv[u] = row_of(ray, id[u], p.N, p.S1, p.S2);
        // (no masking either: every use below is under ok[u], and a select right behind its load is a wait right behind it)
        wv[u] = p.w_row[rowv[u]]; zv[u] = p.z[rowv[u]]; ci[u] = p.cidx_row[rowv[u]];
        dv[u] = 0.f;
        if (p.dots_row) dv[u] = p.dots_row[rowv[u]];
        if (keep) {
          krow[u] = rowv[u];
          kdel[u] = p.delta_row[rowv[u]]; kT[u] = p.T_row[rowv[u]]; ksig[u] = p.sigma[rowv[u]]; kh0h[u] = p.sigma_out[rowv[u] * 16];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CB_KU; ++u) ov[u] = *(const h16x4*)(p.color_out + (size_t)max(ci[u], 0) * 16);   // (used under ci >= 0 only)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CB_KU; ++u) {
        if (!ok[u]) continue;
        const int k = lane + 64 * (i0 + u);
        const float w = wv[u];
        float dw = -p.bg * (gi0 + gi1 + gi2) + zv[u] * gd;
        if (rowmajor_dots) dw += dsem[id[u]];
        dw += dv[u];
        if (ci[u] >= 0) {
          const h16x4 o = ov[u];
          float r0 = 1.f / (1.f + expf(-(float)o[0])), r1 = 1.f / (1.f + expf(-(float)o[1])), r2 = 1.f / (1.f + expf(-(float)o[2]));
          dw += r0 * gi0 + r1 * gi1 + r2 * gi2;
          h16x8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = lo;
          lo[0] = (h16)(w * gi0 * r0 * (1.f - r0)); lo[1] = (h16)(w * gi1 * r1 * (1.f - r1)); lo[2] = (h16)(w * gi2 * r2 * (1.f - r2));
          bad |= !(fabsf((float)lo[0]) <= 65504.f) | !(fabsf((float)lo[1]) <= 65504.f) | !(fabsf((float)lo[2]) <= 65504.f);
          *(h16x8*)(p.d_color_out + (size_t)ci[u] * 16) = lo;
          *(h16x8*)(p.d_color_out + (size_t)ci[u] * 16 + 8) = hi;
        }
        dws[k] = dw; P[k] = dw * w;
      }
    }
    __syncthreads();
    scan_suffix_excl(P, R, S, lane);
    __syncthreads();
    auto finish = [&](int k, size_t row, float delta, float T, float sg, float h0) {
      float om = expf(-delta * p.density_scale * sg);  // 1 - alpha
      float dsig = delta * p.density_scale * om * (dws[k] * T - R[k] / (om + 1e-15f));
      float g = dsig * expf(fminf(fmaxf(h0, -15.f), 15.f));
      bad |= !(fabsf(g) <= 65504.f);
      p.d_h0[row] = g;
    };
    if (keep) {
#pragma unroll
      for (int u = 0; u < CB_KU; ++u) { const int k = lane + 64 * u; if (k < S) finish(k, krow[u], kdel[u], kT[u], ksig[u], (float)kh0h[u]); }
    } else {
      for (int k = lane; k < S; k += 64) {
        int id = p.perm[(size_t)ray * S + k];
        size_t row = row_of(ray, id, p.N, p.S1, p.S2);
        finish(k, row, p.delta_row[row], p.T_row[row], p.sigma[row], (float)p.sigma_out[row * 16]);
      }
    }
    __syncthreads();
  }
  if (p.found_inf && __any(bad) && lane == 0) atomicOr(p.found_inf, 1);
}

extern "C" int aln_composite_bwd(const float* norms, const float* z, const float* sigma, const uint16_t* perm,
                                 const float* w_row, const float* T_row, const float* delta_row, const int32_t* cidx_row,
                                 const void* color_out, const void* logits, const void* feat, const void* sigma_out,
                                 const float* g_image, const float* g_depth, const float* g_sem, const float* g_feat,
                                 int32_t N, int32_t S1, int32_t S2, int32_t C, int32_t Cpad, int32_t D, float bg,
                                 float density_scale, float* d_h0, void* d_color_out, void* d_logits, void* d_feat,
                                 int32_t mask_feat, const float* dots_row, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(norms && z && sigma && perm && w_row && T_row && delta_row && cidx_row && color_out && sigma_out && g_image &&
                  g_depth && d_h0 && d_color_out, "composite_bwd: NULL pointer");
  ALN_REQUIRE((!logits || g_sem) && (!feat || g_feat), "composite_bwd: semantic / feature gradient buffers missing");
  ALN_REQUIRE(Cpad % 8 == 0 && D % 8 == 0, "composite_bwd: Cpad and D must be multiples of 8");
  ALN_REQUIRE(!dots_row || (!logits && !feat && !d_logits && !d_feat), "composite_bwd: dots_row replaces the logits / feat rows (pass those as NULL)");
  if (N <= 0) return 0;
  CompBwd p{norms, z, sigma, perm, w_row, T_row, delta_row, cidx_row, (const h16*)color_out, (const h16*)logits,
            (const h16*)feat, (const h16*)sigma_out, g_image, g_depth, g_sem, g_feat, N, S1, S2, C, Cpad, D, mask_feat, bg,
            density_scale, d_h0, (h16*)d_color_out, (h16*)d_logits, (h16*)d_feat, found_inf, dots_row};
  int S = S1 + S2;
  size_t lds = (size_t)(4 * S + Cpad + D) * sizeof(float);
  hipLaunchKernelGGL(k_composite_bwd, dim3(N < 65535 ? N : 65535), dim3(64), lds, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("composite_bwd");
  return 0;
}
