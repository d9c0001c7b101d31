// Occupancy-grid ray marching (the `cuda_ray` path: hooks at autolabel/trainer.py:21-23,34-36,176; the reference always
// passes cuda_ray=False, autolabel/model_utils.py:72, so this is SURVEY.md 8(f) N1 -- beyond reference parity).
// Spec: oracle/march_oracle.py (own spec after upstream torch-ngp raymarching / instant-ngp):
//
//  * density grid: G^3 floats over [-bound, bound]^3 (one level; G = 128), cell (ix, iy, iz) -> ix + G (iy + G iz); -1 marks
//    cells no training camera sees (mark_untrained_grid); bitfield: one bit per cell, set where grid > threshold.
//  * update (update_extra_state, every 16 steps): one jittered point per cell -> density head -> grid = max(grid * decay,
//    sigma * density_scale); mean over the cells >= 0; threshold = min(mean, density_thresh).
//  * marching: steps t_i = near + (i + u) * dt, i = 0 .. ceil((far - near) / dt) - 1, dt = 2 sqrt(3) bound / max_steps, u one
//    uniform per ray (perturb) or 0.5.  K = number of steps whose cell bit is set.  The ray gets exactly S sample rows (the
//    fixed [N, S] layout every downstream kernel uses -- nothing in the step has a data-dependent size, so it stays
//    capturable in a hipGraph): K <= S: the K occupied steps, delta = dt, then S - K padding rows (z of the last step, delta
//    0 => weight 0).  K > S: the occupied steps are subsampled evenly -- occupied step r is kept iff it is the first with
//    floor(r S / K) = j, it lands in row j -- and delta = dt K / S, i.e. a coarser step over the same occupied length.
//
// MI355X mapping: one 64-lane wave per ray; the wave tests 64 consecutive steps per iteration (the 256 KB bitfield is
// L2-resident), `__ballot` + popcount give every occupied step its rank r without any atomic; two sweeps (count, emit).
#include "common.h"
#include <math.h>

struct MarchArgs {
  const float* ro; const float* rd; int N, S; float bound, min_near;
  const uint32_t* bits; int G, max_steps; int perturb; uint32_t seed, step; const uint32_t* step_dev; const float* noise;
  float* nears; float* fars; float* z; float* delta; int* counts;
};

__device__ inline void march_ray_aabb(const float* o3, const float* d3, float bound, float min_near, float& near, float& far) {
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

__device__ inline int grid_cell(const float* o, const float* d, float t, float bound, int G) {
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float p = __fadd_rn(o[k], __fmul_rn(d[k], t));
    p = fminf(fmaxf(p, -bound), bound);
    float u = __fmul_rn(__fdiv_rn(__fadd_rn(p, bound), __fmul_rn(2.0f, bound)), (float)G);
    int i = (int)floorf(u);
    c[k] = i < 0 ? 0 : (i > G - 1 ? G - 1 : i);
  }
  return c[0] + G * (c[1] + G * c[2]);
}

__global__ __launch_bounds__(256) void k_march_rays(MarchArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const float* o = a.ro + 3 * (size_t)ray;
  const float* d = a.rd + 3 * (size_t)ray;
  float near, far;
  march_ray_aabb(o, d, a.bound, a.min_near, near, far);
  const float dt = __fdiv_rn(__fmul_rn(3.4641016151377544f, a.bound), (float)a.max_steps);
  float u = 0.5f;
  if (a.perturb) {
    const uint32_t key = aln_rand_key(a.seed, ALN_STREAM_PERTURB, a.step + (a.step_dev ? *a.step_dev : 0u));
    u = a.noise ? a.noise[ray] : aln_rand_uniform(key, (uint32_t)ray);
  }
  int n_steps = (int)ceilf(__fdiv_rn(__fsub_rn(far, near), dt));
  n_steps = n_steps < 0 ? 0 : (n_steps > a.max_steps ? a.max_steps : n_steps);
  auto occupied = [&](int i, float& t) {
    t = __fadd_rn(near, __fmul_rn(__fadd_rn((float)i, u), dt));
    if (i >= n_steps) return false;
    const int c = grid_cell(o, d, t, a.bound, a.G);
    return ((a.bits[c >> 5] >> (c & 31)) & 1u) != 0u;
  };
  // sweep 1: K
  int K = 0;
  for (int i0 = 0; i0 < n_steps; i0 += 64) { float t; K += __popcll(__ballot(occupied(i0 + lane, t))); }
  // sweep 2: emit
  const int S = a.S;
  float* zr = a.z + (size_t)ray * S;
  float* dr = a.delta + (size_t)ray * S;
  const float dl = K > S ? __fdiv_rn(__fmul_rn(dt, (float)K), (float)S) : dt;
  int r0 = 0; float t_last = near;
  for (int i0 = 0; i0 < n_steps; i0 += 64) {
    float t;
    const bool occ = occupied(i0 + lane, t);
    const unsigned long long m = __ballot(occ);
    if (occ) {
      const int r = r0 + __popcll(m & ((1ull << lane) - 1ull));
      if (K <= S) { zr[r] = t; dr[r] = dl; }
      else {
        const int j = (int)(((long long)r * S) / K);
        const bool first = r == 0 || (int)(((long long)(r - 1) * S) / K) != j;
        if (first) { zr[j] = t; dr[j] = dl; }
      }
    }
    if (m) { const int hi = 63 - __clzll(m); t_last = __shfl(t, hi); }
    r0 += __popcll(m);
  }
  const int kept = K < S ? K : S;
  for (int j = kept + lane; j < S; j += 64) { zr[j] = t_last; dr[j] = 0.f; }
  if (lane == 0) {
    a.nears[ray] = near; a.fars[ray] = far;
    if (a.counts) a.counts[ray] = K;
  }
}

extern "C" int aln_march_rays(const float* rays_o, const float* rays_d, int32_t N, int32_t S, float bound, float min_near,
                              const uint32_t* bitfield, int32_t G, int32_t max_steps, int32_t perturb, uint32_t seed,
                              uint32_t step, const uint32_t* step_dev, const float* noise, float* nears, float* fars, float* z,
                              float* delta, int32_t* counts, void* stream) {
  ALN_REQUIRE(rays_o && rays_d && bitfield && nears && fars && z && delta, "march_rays: NULL pointer");
  ALN_REQUIRE(S > 0 && G > 0 && G <= 1024 && max_steps > 0 && max_steps <= 65536, "march_rays: S=%d G=%d max_steps=%d", S, G, max_steps);
  if (N <= 0) return 0;
  MarchArgs a{rays_o, rays_d, N, S, bound, min_near, bitfield, G, max_steps, perturb, seed, step, step_dev, noise,
              nears, fars, z, delta, counts};
  hipLaunchKernelGGL(k_march_rays, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("march_rays");
  return 0;
}

// ---------------------------------------------------------------- density grid maintenance
// one jittered point per cell (cell centre +- half a cell)
__global__ void k_grid_points(int G, float bound, uint32_t seed, uint32_t step, const uint32_t* __restrict__ step_dev,
                              const float* __restrict__ noise, float* __restrict__ xyz) {
  const uint32_t key = aln_rand_key(seed, ALN_STREAM_PERTURB, step + (step_dev ? *step_dev : 0u));
  const uint32_t n = (uint32_t)G * G * G, g = (uint32_t)G;     // G <= 1024: 32-bit index arithmetic
  for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    const uint32_t ix = c % g, r = c / g, iy = r % g, iz = r / g;
    const float cf[3] = {(float)ix, (float)iy, (float)iz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float u = noise ? noise[3 * (size_t)c + k] : aln_rand_uniform(key, 3u * c + (uint32_t)k);
      const float p = __fmul_rn(__fdiv_rn(__fadd_rn(cf[k], u), (float)G), __fmul_rn(2.0f, bound));
      xyz[3 * (size_t)c + k] = __fsub_rn(p, bound);
    }
  }
}

extern "C" int aln_grid_points(int32_t G, float bound, uint32_t seed, uint32_t step, const uint32_t* step_dev, const float* noise,
                               float* xyz, void* stream) {
  ALN_REQUIRE(G > 0 && G <= 1024 && xyz, "grid_points: bad arguments");
  hipLaunchKernelGGL(k_grid_points, dim3(aln_grid_for((int64_t)G * G * G, 256, 8192)), dim3(256), 0, (hipStream_t)stream, G, bound,
                     seed, step, step_dev, noise, xyz);
  ALN_CHECK_LAUNCH("grid_points");
  return 0;
}

// grid = max(grid * decay, sigma * density_scale) on cells >= 0; stats[0] += sum (fixed point, 2^-16), stats[1] += count
// (zero them first).  Integer accumulation: the mean -- and with it the bitfield of cells sitting at the threshold -- does not
// depend on the order in which the waves arrive.  sigma == NULL: statistics of the grid as it is.
#define GRID_SUM_SCALE 65536.0
__global__ void k_grid_ema(float* grid, const float* sigma, size_t n, float decay, float density_scale,
                           unsigned long long* __restrict__ stats) {
  float s = 0.f, cnt = 0.f;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    float g = grid[c];
    if (g >= 0.f) {
      if (sigma) { g = fmaxf(__fmul_rn(g, decay), __fmul_rn(sigma[c], density_scale)); grid[c] = g; }
      s += g; cnt += 1.f;
    }
  }
  s = wave_sum(s); cnt = wave_sum(cnt);   // fixed lane order within the wave; the waves meet in integer adds
  if ((threadIdx.x & 63) == 0 && cnt > 0.f) {
    atomicAdd(&stats[0], (unsigned long long)((double)fminf(s, 1e30f) * GRID_SUM_SCALE));
    atomicAdd(&stats[1], (unsigned long long)cnt);
  }
}

// bit c = grid[c] > min(mean, thresh)   (32 cells per thread-word: a wave writes 256 contiguous bytes)
__global__ void k_grid_bits(const float* __restrict__ grid, size_t n, const unsigned long long* __restrict__ stats, float thresh,
                            uint32_t* __restrict__ bits, int* __restrict__ n_set) {
  const float mean = stats[1] ? (float)((double)stats[0] / GRID_SUM_SCALE / (double)stats[1]) : 0.f;
  const float th = fminf(mean, thresh);
  int local = 0;
  const size_t nw = (n + 31) / 32;
  for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < nw; w += (size_t)gridDim.x * blockDim.x) {
    uint32_t word = 0u;
    for (int b = 0; b < 32; ++b) {
      const size_t c = w * 32 + b;
      if (c < n && grid[c] > th) word |= 1u << b;
    }
    bits[w] = word;
    local += __popc(word);
  }
  if (n_set) {
    float f = wave_sum((float)local);
    if ((threadIdx.x & 63) == 0 && f > 0.f) atomicAdd(n_set, (int)f);
  }
}

extern "C" int aln_grid_update(float* grid, const float* sigma, int32_t G, float decay, float density_scale, float thresh,
                               void* stats /*16 bytes of scratch*/, uint32_t* bitfield, int32_t* n_set /*[1] or NULL*/, void* stream) {
  ALN_REQUIRE(grid && stats && bitfield && G > 0, "grid_update: bad arguments");
  const size_t n = (size_t)G * G * G;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) { aln_set_error("grid_update: memset failed"); return -2; }
  if (n_set && hipMemsetAsync(n_set, 0, sizeof(int), s) != hipSuccess) { aln_set_error("grid_update: memset failed"); return -2; }
  // sigma == NULL: statistics of the grid as it is (after mark_untrained_grid)
  hipLaunchKernelGGL(k_grid_ema, dim3(aln_grid_for((int64_t)n, 256, 2048)), dim3(256), 0, s, grid, sigma, n, decay, density_scale,
                     (unsigned long long*)stats);
  ALN_CHECK_LAUNCH("grid_ema");
  hipLaunchKernelGGL(k_grid_bits, dim3(aln_grid_for((int64_t)((n + 31) / 32), 256, 2048)), dim3(256), 0, s, grid, n,
                     (const unsigned long long*)stats, thresh, bitfield, n_set);
  ALN_CHECK_LAUNCH("grid_bits");
  return 0;
}

// n_set = number of set bits of a bitfield that arrived with a checkpoint (the bits themselves are kept as stored)
__global__ void k_bitfield_count(const uint32_t* __restrict__ bits, size_t nw, int* __restrict__ n_set) {
  int local = 0;
  for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < nw; w += (size_t)gridDim.x * blockDim.x) local += __popc(bits[w]);
  const float f = wave_sum((float)local);
  if ((threadIdx.x & 63) == 0 && f > 0.f) atomicAdd(n_set, (int)f);
}
extern "C" int aln_bitfield_count(const uint32_t* bitfield, int64_t n_words, int32_t* n_set, void* stream) {
  ALN_REQUIRE(bitfield && n_set && n_words >= 0, "bitfield_count: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(n_set, 0, sizeof(int), s) != hipSuccess) { aln_set_error("bitfield_count: memset failed"); return -2; }
  if (n_words == 0) return 0;
  hipLaunchKernelGGL(k_bitfield_count, dim3(aln_grid_for(n_words, 256, 2048)), dim3(256), 0, s, bitfield, (size_t)n_words, n_set);
  ALN_CHECK_LAUNCH("bitfield_count");
  return 0;
}

// mark_untrained_grid: a cell stays trainable (>= 0) iff at least one of its `sub`^3 sub-points projects into some camera
// image in front of the camera (OpenCV pinhole: x right, y down, z forward; T_CW = world -> camera, row-major 4x4, in the
// renderer's world frame).  Unseen cells get -1 and can never become occupied.
__global__ void k_mark_untrained(float* __restrict__ grid, int G, float bound, const float* __restrict__ T_CW, int n_poses,
                                 float fx, float fy, float cx, float cy, float w, float h, float z_near, int sub) {
  const size_t n = (size_t)G * G * G;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    const int ci[3] = {(int)(c % G), (int)((c / G) % G), (int)(c / ((size_t)G * G))};
    bool seen = false;
    for (int s = 0; s < sub * sub * sub && !seen; ++s) {
      const int si[3] = {s % sub, (s / sub) % sub, s / (sub * sub)};
      float p[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = ((float)ci[k] + ((float)si[k] + 0.5f) / (float)sub) / (float)G * 2.0f * bound - bound;
      for (int q = 0; q < n_poses && !seen; ++q) {
        const float* T = T_CW + 16 * (size_t)q;
        const float zc = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
        if (zc <= z_near) continue;
        const float xc = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
        const float yc = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
        const float px = fx * xc / zc + cx, py = fy * yc / zc + cy;
        seen = px >= 0.f && px <= w && py >= 0.f && py <= h;
      }
    }
    grid[c] = seen ? fmaxf(grid[c], 0.f) : -1.f;
  }
}

extern "C" int aln_mark_untrained_grid(float* grid, int32_t G, float bound, const float* T_CW, int32_t n_poses, float fx,
                                       float fy, float cx, float cy, float w, float h, float z_near, int32_t sub, void* stream) {
  ALN_REQUIRE(grid && T_CW && G > 0 && n_poses > 0 && sub > 0 && sub <= 4, "mark_untrained_grid: bad arguments");
  hipLaunchKernelGGL(k_mark_untrained, dim3(aln_grid_for((int64_t)G * G * G, 256, 8192)), dim3(256), 0, (hipStream_t)stream, grid, G,
                     bound, T_CW, n_poses, fx, fy, cx, cy, w, h, z_near, sub);
  ALN_CHECK_LAUNCH("mark_untrained");
  return 0;
}
