// Glue between the sigma MLP and the color / semantic heads: live-sample compaction, head input
// assembly, and the matching gradient assembly.  Mirrors the tensor plumbing of
// autolabel/models.py:175-188 (density: sigma = trunc_exp(h0), geo_feat = h[1:]), :190-220 (color: boolean-mask
// gather, SH(dir) ++ geo_feat), :248-256 (semantic: cat[relu(f), geo_feat]).
#include "common.h"
#include <math.h>

// sigma[row] = exp(h0)  (torch-ngp trunc_exp forward, fp32)
__global__ void k_sigma_act(const h16* __restrict__ sigma_out, int rows, float* __restrict__ sigma) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    sigma[r] = expf((float)sigma_out[(size_t)r * 16]);
}
extern "C" int aln_sigma_act(const void* sigma_out, int32_t rows, float* sigma, void* stream) {
  ALN_REQUIRE(sigma_out && sigma, "sigma_act: NULL pointer");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_sigma_act, dim3(aln_grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, (const h16*)sigma_out,
                     rows, sigma);
  ALN_CHECK_LAUNCH("sigma_act");
  return 0;
}

// One input row of the colour head: color_in[ci] = [SH16(dir(ray(row))), geo_feat[row] (G), 1...], width in_pad, whole 16-byte
// chunks in and out (a row is 64 B for in_pad = 32).  Shared by k_build_color_in and the compaction's second pass.
struct ColorInSrc { const float* rd; const float* dirs; int N, S1, S2; const h16* sigma_out; int G, in_pad; h16* cin; };
// (Round 6, measured and rejected: FOUR lanes per row, lane q storing chunk q -- 1 KB of contiguous rows per wave instruction instead of
//  four 16-byte stores at a 64-byte stride: 27 -> 35 us standalone, 37 -> 87 us inside the compaction.  The kernel moves 100 MB per
//  million live rows at 3.7 TB/s as it is; the stores were not the problem, the four-fold arithmetic is one.)
__device__ inline void build_color_row(const ColorInSrc& s, int row, int ci) {
  const float* d;
  if (s.dirs) d = s.dirs + 3 * (size_t)row;
  else { int ray = row < s.N * s.S1 ? row / s.S1 : (row - s.N * s.S1) / s.S2; d = s.rd + 3 * (size_t)ray; }
  float sh[16];
  sh4_of_dir(d, sh);
  const h16x8 lo = *(const h16x8*)(s.sigma_out + (size_t)row * 16), hi = *(const h16x8*)(s.sigma_out + (size_t)row * 16 + 8);
  h16x8 c0, c1, c2, c3;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c0[j] = (h16)sh[j]; c1[j] = (h16)sh[8 + j]; }
#pragma unroll
  for (int j = 0; j < 8; ++j) {   // geo_feat[g] = sigma_out[row][1 + g], ones from G on
    h16 a = (j < 7) ? lo[j + 1] : hi[0], b = (j < 7) ? hi[j + 1] : (h16)1.0f;
    c2[j] = (j < s.G) ? a : (h16)1.0f; c3[j] = (8 + j < s.G) ? b : (h16)1.0f;
  }
  h16* o = s.cin + (size_t)ci * s.in_pad;
  *(h16x8*)o = c0; *(h16x8*)(o + 8) = c1; *(h16x8*)(o + 16) = c2;
  if (s.in_pad >= 32) *(h16x8*)(o + 24) = c3;
  for (int j = 32; j < s.in_pad; ++j) o[j] = (h16)1.0f;
}

// live = w > thresh (renderer: mask = weights > 1e-4).  Ballot compaction in ROW ORDER: pass 1 counts the live rows of every
// 1024-row chunk, pass 2 re-derives the bits, takes the sum of the earlier chunks' counts as its base (<= 2048 chunks at 2^21
// rows: a few coalesced reads per block) and writes.  The compact order is a pure function of w_row -- the color head sees its rows,
// and forms its weight-gradient partial sums, in the same order every run (round 2 handed out chunk bases with a returning
// atomic: block arrival order).
#ifndef COMPACT_ITERS
#define COMPACT_ITERS 4    // 64-row groups per wave: 1024-row chunks (round 6; 16 -- 64 blocks for a 1024-ray batch, each wave building its colour rows
                           // group after group -- cost 20 us there)
#endif
#define COMPACT_CHUNK (256 * COMPACT_ITERS)
__global__ __launch_bounds__(256) void k_compact_count(const float* __restrict__ w_row, int rows, float thresh, int* __restrict__ chunk_cnt) {
  __shared__ int s_cnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunks = (rows + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int r0 = ch * COMPACT_CHUNK + wave * 64 * COMPACT_ITERS;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < COMPACT_ITERS; ++i) {
      const int r = r0 + i * 64 + lane;
      const float wv = w_row[min(r, rows - 1)];   // (unconditional: behind a load in a branch hipcc waits for each of the sixteen in turn)
      cnt += __popcll(__ballot(r < rows && wv > thresh));
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[ch] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_compact_live(const float* __restrict__ w_row, int rows, float thresh, const int* __restrict__ chunk_cnt,
                                                     int* __restrict__ n_live, int* __restrict__ live_idx, int* __restrict__ cidx_row, ColorInSrc col) {
  __shared__ int s_cnt[4], s_part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunks = (rows + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    // base = number of live rows in chunks [0, ch): integer sum, any order gives the same value
    int part = 0;
    for (int q = threadIdx.x; q < ch; q += 256) part += chunk_cnt[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const int r0 = ch * COMPACT_CHUNK + wave * 64 * COMPACT_ITERS;  // each wave owns a contiguous span of the chunk
    unsigned bits = 0; int cnt = 0;
#pragma unroll
    for (int i = 0; i < COMPACT_ITERS; ++i) {
      int r = r0 + i * 64 + lane;
      const float wv = w_row[min(r, rows - 1)];
      bool live = r < rows && wv > thresh;
      bits |= (unsigned)live << i;
      cnt += __popcll(__ballot(live));
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_part[wave] = part; }
    __syncthreads();
    int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    if (ch == nchunks - 1 && threadIdx.x == 0) *n_live = s_part[0] + s_part[1] + s_part[2] + s_part[3] + s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
#pragma unroll
    for (int i = 0; i < COMPACT_ITERS; ++i) {
      int r = r0 + i * 64 + lane;
      bool live = (bits >> i) & 1u;
      unsigned long long m = __ballot(live);
      if (r < rows) {
        int ci = live ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        cidx_row[r] = ci;
        if (live) {
          live_idx[ci] = r;
          if (col.cin) build_color_row(col, r, ci);   // (round 6: the colour head's input row, built where its place becomes known)
        }
      }
      off += __popcll(m);
    }
    __syncthreads();
  }
}
// chunk_ws: caller-owned scratch of aln_compact_live_ws_ints(rows) int32 (the per-chunk counts between the two passes)
extern "C" int32_t aln_compact_live_ws_ints(int32_t rows) { return rows > 0 ? (rows + COMPACT_CHUNK - 1) / COMPACT_CHUNK : 0; }
static int compact_launch(const float* w_row, int32_t rows, float thresh, int32_t* n_live, int32_t* live_idx, int32_t* cidx_row,
                          int32_t* chunk_ws, ColorInSrc col, void* stream) {
  ALN_REQUIRE(w_row && n_live && live_idx && cidx_row && chunk_ws, "compact_live: NULL pointer");
  if (rows <= 0) { hipMemsetAsync(n_live, 0, sizeof(int), (hipStream_t)stream); return 0; }
  const int nchunks = (rows + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
  hipLaunchKernelGGL(k_compact_count, dim3(nchunks < 1024 ? nchunks : 1024), dim3(256), 0, (hipStream_t)stream, w_row, rows, thresh, chunk_ws);
  ALN_CHECK_LAUNCH("compact_count");
  hipLaunchKernelGGL(k_compact_live, dim3(nchunks < 1024 ? nchunks : 1024), dim3(256), 0, (hipStream_t)stream, w_row, rows, thresh,
                     (const int*)chunk_ws, n_live, live_idx, cidx_row, col);
  ALN_CHECK_LAUNCH("compact_live");
  return 0;
}
extern "C" int aln_compact_live(const float* w_row, int32_t rows, float thresh, int32_t* n_live, int32_t* live_idx,
                                int32_t* cidx_row, int32_t* chunk_ws, void* stream) {
  return compact_launch(w_row, rows, thresh, n_live, live_idx, cidx_row, chunk_ws, ColorInSrc{}, stream);
}
// the same, and color_in[cidx_row[r]] (aln_build_color_in's rows, bit for bit) for every live row r out of the second pass: no
// k_build_color_in launch, no second read of live_idx (the training step, round 6)
extern "C" int aln_compact_live_color_in(const float* w_row, int32_t rows, float thresh, int32_t* n_live, int32_t* live_idx,
                                         int32_t* cidx_row, int32_t* chunk_ws, const float* rays_d, const float* dirs, int32_t N,
                                         int32_t S1, int32_t S2, const void* sigma_out, int32_t G, int32_t in_pad, void* color_in,
                                         void* stream) {
  ALN_REQUIRE((rays_d || dirs) && sigma_out && color_in, "compact_live_color_in: NULL pointer");
  ALN_REQUIRE(G + 1 <= 16 && 16 + G <= in_pad && in_pad % 8 == 0 && (G <= 8 || in_pad >= 32), "compact_live_color_in: geo_feat_dim %d unsupported", G);
  return compact_launch(w_row, rows, thresh, n_live, live_idx, cidx_row, chunk_ws,
                        ColorInSrc{rays_d, dirs, N, S1, S2, (const h16*)sigma_out, G, in_pad, (h16*)color_in}, stream);
}

__global__ void k_sh4(const float* __restrict__ dirs, int rows, int pitch, h16* __restrict__ out) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    float sh[16];
    sh4_of_dir(dirs + 3 * (size_t)r, sh);
    h16* o = out + (size_t)r * pitch;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = (h16)sh[j];
  }
}
extern "C" int aln_sh4(const float* dirs, int32_t rows, int32_t out_pitch, void* out, void* stream) {
  ALN_REQUIRE(dirs && out && out_pitch >= 16, "sh4: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_sh4, dim3(aln_grid_for(rows, 256)), dim3(256), 0, (hipStream_t)stream, dirs, rows, out_pitch, (h16*)out);
  ALN_CHECK_LAUNCH("sh4");
  return 0;
}

__global__ void k_build_color_in(const int* __restrict__ live_idx, const int* __restrict__ n_live, int max_rows, ColorInSrc src) {
  int n = live_idx ? min(*n_live, max_rows) : max_rows;   // one thread per row
  for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < n; ci += gridDim.x * blockDim.x) build_color_row(src, live_idx ? live_idx[ci] : ci, ci);
}
extern "C" int aln_build_color_in(const int32_t* live_idx, const int32_t* n_live, int32_t max_rows, const float* rays_d,
                                  const float* dirs, int32_t N, int32_t S1, int32_t S2, const void* sigma_out, int32_t G,
                                  int32_t in_pad, void* color_in, void* stream) {
  ALN_REQUIRE((rays_d || dirs) && sigma_out && color_in && (!live_idx || n_live), "build_color_in: NULL pointer");
  ALN_REQUIRE(G + 1 <= 16 && 16 + G <= in_pad && in_pad % 8 == 0 && (G <= 8 || in_pad >= 32), "build_color_in: geo_feat_dim %d unsupported", G);
  if (max_rows <= 0) return 0;
  hipLaunchKernelGGL(k_build_color_in, dim3(aln_grid_for(max_rows, 256)), dim3(256), 0, (hipStream_t)stream, live_idx, n_live,
                     max_rows, ColorInSrc{rays_d, dirs, N, S1, S2, (const h16*)sigma_out, G, in_pad, (h16*)color_in});
  ALN_CHECK_LAUNCH("build_color_in");
  return 0;
}

// semf_in[row] = [geo_feat (G), 1...] (16 wide)
__global__ void k_build_semf_in(const h16* __restrict__ sigma_out, int rows, int G, int in_pad, h16* __restrict__ o) {
  size_t total = (size_t)rows * in_pad;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    size_t r = t / in_pad; int j = (int)(t % in_pad);
    o[t] = j < G ? sigma_out[r * 16 + 1 + j] : (h16)1.0f;
  }
}
// semo_in[row] = [relu(f) (D), geo_feat (G), 1...]
__global__ void k_build_semo_in(const h16* __restrict__ f, const h16* __restrict__ sigma_out, int rows, int D, int G,
                                int in_pad, h16* __restrict__ o) {
  // one 8-feature chunk per thread (D % 8 == 0, in_pad % 8 == 0): relu(f) chunks, then [geo_feat, 1...] chunks
  const int nch = in_pad / 8;
  size_t total = (size_t)rows * nch;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    size_t r = t / nch; int c0 = 8 * (int)(t % nch);
    h16x8 v;
    if (c0 < D) {
      v = *(const h16x8*)(f + r * D + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) if ((float)v[j] < 0.f) v[j] = (h16)0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { int g = c0 - D + j; v[j] = g < G ? sigma_out[r * 16 + 1 + g] : (h16)1.0f; }
    }
    *(h16x8*)(o + r * in_pad + c0) = v;
  }
}
extern "C" int aln_build_sem_in(const void* sigma_out, const void* f, int32_t rows, int32_t D, int32_t G, int32_t semf_in_pad,
                                int32_t semo_in_pad, void* semf_in, void* semo_in, void* stream) {
  ALN_REQUIRE(sigma_out, "build_sem_in: NULL pointer");
  if (rows <= 0) return 0;
  if (semf_in) {
    hipLaunchKernelGGL(k_build_semf_in, dim3(aln_grid_for((int64_t)rows * semf_in_pad, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const h16*)sigma_out, rows, G, semf_in_pad, (h16*)semf_in);
    ALN_CHECK_LAUNCH("build_semf_in");
  }
  if (semo_in) {
    ALN_REQUIRE(f, "build_sem_in: f is NULL");
    ALN_REQUIRE(D % 8 == 0 && semo_in_pad % 8 == 0, "build_sem_in: D and semo_in_pad must be multiples of 8");
    hipLaunchKernelGGL(k_build_semo_in, dim3(aln_grid_for((int64_t)rows * (semo_in_pad / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const h16*)f, (const h16*)sigma_out, rows, D, G, semo_in_pad, (h16*)semo_in);
    ALN_CHECK_LAUNCH("build_semo_in");
  }
  return 0;
}

// d_f_total[row][d] = d_feat[row][d] + (f > 0 ? d_semo_in[row][d] : 0)      (in place into d_feat)
__global__ void k_assemble_dsemf_out(h16* __restrict__ d_feat, const h16* __restrict__ f, const h16* __restrict__ d_semo_in,
                                     int rows, int D, int semo_in_pad, int* __restrict__ found_inf) {
  // d_feat += relu'(f) * d_semo_in[:, :D], 8 features per thread
  const int nch = D / 8;
  size_t total = (size_t)rows * nch;
  bool bad = false;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    size_t r = t / nch; int c0 = 8 * (int)(t % nch);
    h16x8 g = *(const h16x8*)(d_feat + r * D + c0);
    const h16x8 fv = *(const h16x8*)(f + r * D + c0), dv = *(const h16x8*)(d_semo_in + r * semo_in_pad + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = (float)g[j];
      if ((float)fv[j] > 0.f) x += (float)dv[j];
      g[j] = (h16)x; bad |= !(fabsf((float)g[j]) <= 65504.f);
    }
    *(h16x8*)(d_feat + r * D + c0) = g;
  }
  if (found_inf && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}
// d_sigma_out[row] = [d_h0, d_geo(semf_in) + d_geo(semo_in) + d_geo(color_in)] (16 wide)
__global__ void k_assemble_dsigma_out(const float* __restrict__ d_h0, const h16* __restrict__ d_semf_in, int semf_in_pad,
                                      const h16* __restrict__ d_semo_in, int semo_in_pad, int D, const h16* __restrict__ d_color_in,
                                      int color_in_pad, const int* __restrict__ cidx_row, int rows, int G,
                                      h16* __restrict__ d_sigma_out, int* __restrict__ found_inf) {
  // one thread per row: the geo_feat gradients of the three consumers arrive as whole 16-byte chunks
  bool bad = false;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    float g[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) g[j] = 0.f;
    if (d_semf_in) {
      const h16* pf = d_semf_in + (size_t)r * semf_in_pad;
      const h16x8 f0 = *(const h16x8*)pf, f1 = *(const h16x8*)(pf + 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { g[j] = (float)f0[j]; g[8 + j] = (float)f1[j]; }
      if (d_semo_in) {   // (NULL: the fused semantic backward has already added the skip-connection part into d_semf_in)
        const h16* po = d_semo_in + (size_t)r * semo_in_pad + D;
        const h16x8 o0 = *(const h16x8*)po, o1 = *(const h16x8*)(po + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] += (float)o0[j]; g[8 + j] += (float)o1[j]; }
      }
    }
    const int ci = cidx_row ? cidx_row[r] : r;
    if (ci >= 0) {
      const h16* pc = d_color_in + (size_t)ci * color_in_pad + 16;
      const h16x8 c0 = *(const h16x8*)pc, c1 = *(const h16x8*)(pc + 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { g[j] += (float)c0[j]; g[8 + j] += (float)c1[j]; }
    }
    h16x8 lo, hi;
    lo[0] = (h16)d_h0[r];
#pragma unroll
    for (int j = 1; j < 16; ++j) { h16 v = (j <= G) ? (h16)g[j - 1] : (h16)0.f; if (j < 8) lo[j] = v; else hi[j - 8] = v; }
#pragma unroll
    for (int j = 0; j < 8; ++j) bad |= !(fabsf((float)lo[j]) <= 65504.f) | !(fabsf((float)hi[j]) <= 65504.f);
    *(h16x8*)(d_sigma_out + (size_t)r * 16) = lo; *(h16x8*)(d_sigma_out + (size_t)r * 16 + 8) = hi;
  }
  if (found_inf && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}
extern "C" int aln_assemble_grads(const float* d_h0, const void* d_semf_in, int32_t semf_in_pad, const void* d_semo_in,
                                  int32_t semo_in_pad, int32_t D, const void* d_color_in, int32_t color_in_pad,
                                  const int32_t* cidx_row, int32_t rows, int32_t G, void* d_sigma_out, int32_t* found_inf,
                                  void* stream) {
  ALN_REQUIRE(d_h0 && d_color_in && d_sigma_out && (d_semf_in || !d_semo_in), "assemble_grads: NULL pointer");
  if (rows <= 0) return 0;
  ALN_REQUIRE(G <= 15 && color_in_pad >= 32 && color_in_pad % 8 == 0 && (!d_semf_in || (semf_in_pad == 16 && (!d_semo_in || (D % 8 == 0 && semo_in_pad >= D + 16)))),
              "assemble_grads: unsupported widths");
  hipLaunchKernelGGL(k_assemble_dsigma_out, dim3(aln_grid_for((int64_t)rows, 256)), dim3(256), 0, (hipStream_t)stream, d_h0,
                     (const h16*)d_semf_in, semf_in_pad, (const h16*)d_semo_in, semo_in_pad, D, (const h16*)d_color_in,
                     color_in_pad, cidx_row, rows, G, (h16*)d_sigma_out, found_inf);
  ALN_CHECK_LAUNCH("assemble_dsigma_out");
  return 0;
}
extern "C" int aln_assemble_dsemf_out(void* d_feat, const void* f, const void* d_semo_in, int32_t rows, int32_t D,
                                      int32_t semo_in_pad, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(d_feat && f && d_semo_in, "assemble_dsemf_out: NULL pointer");
  if (rows <= 0) return 0;
  ALN_REQUIRE(D % 8 == 0 && semo_in_pad % 8 == 0, "assemble_dsemf_out: D and semo_in_pad must be multiples of 8");
  hipLaunchKernelGGL(k_assemble_dsemf_out, dim3(aln_grid_for((int64_t)rows * (D / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (h16*)d_feat, (const h16*)f, (const h16*)d_semo_in, rows, D, semo_in_pad, found_inf);
  ALN_CHECK_LAUNCH("assemble_dsemf_out");
  return 0;
}

// ---- open-vocabulary prompt comparison (autolabel/evaluation.py:304-323, 426-443): out[r] = argmax_c <f_r, t_c> in fp32, first
// maximum wins, all-zero rows -> 0 (the reference divides by the norm first: NaN rows argmax to 0 there).  One wave per row at a
// time: the row is read coalesced, the prompt matrix [C, D] stays in L1 / L2 (tens of prompts).
__global__ __launch_bounds__(256) void k_similarity_argmax(const float* __restrict__ f, int n, int D, const float* __restrict__ t, int C,
                                                          long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
    const float* fr = f + (size_t)r * D;
    float best = -INFINITY; int arg = 0; bool any = false;
    for (int d = lane; d < D; d += 64) any |= fr[d] != 0.f;
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
      for (int d = lane; d < D; d += 64) acc = fmaf(fr[d], t[(size_t)c * D + d], acc);
      acc = wave_sum(acc);
      if (acc > best) { best = acc; arg = c; }
    }
    if (!__any(any)) arg = 0;
    if (lane == 0) out[r] = arg;
  }
}
extern "C" int aln_similarity_argmax(const float* features, int32_t n, int32_t D, const float* text, int32_t C, int64_t* out, void* stream) {
  ALN_REQUIRE(features && text && out && n >= 0 && D > 0 && C > 0, "similarity_argmax: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_similarity_argmax, dim3(aln_grid_for((int64_t)n, 4, 256 * 16)), dim3(256), 0, (hipStream_t)stream, features, n, D, text, C,
                     (long long*)out);
  ALN_CHECK_LAUNCH("similarity_argmax");
  return 0;
}
