// Device-resident ray generation / batch assembly.
// Replaces the host numpy+numba path autolabel/dataset.py:17-37 (_compute_direction), :182-242
// (BaseDataset._next_train), :244-266 (_get_test).  Spec: oracle/raygen_oracle.py, pinned to the
// reference's own outputs (tests/golden/raygen_f*.npz).
#include "common.h"
#include <math.h>

// dataset.py:17-37.  The pinhole division runs in float64 (fx..cy are float64 scalars) and is rounded once
// to float32; norm, normalisation and rotation are float32, unfused, in the order of the oracle.
__device__ inline void pixel_direction(const float* __restrict__ R, int64_t idx, int w, double fx, double fy, double cx,
                                       double cy, bool randomize, float jx, float jy, float* dir, float* norm_out) {
  int64_t xi = idx % w;
  float xs = (float)xi, ys = (float)((idx - xi) / w);
  if (randomize) { xs = __fadd_rn(xs, jx); ys = __fadd_rn(ys, jy); }
  else { xs = __fadd_rn(xs, 0.5f); ys = __fadd_rn(ys, 0.5f); }
  float d0 = (float)(((double)xs - cx) / fx), d1 = (float)(((double)ys - cy) / fy), d2 = 1.0f;
  // v_sqrt_f32 is 1-ulp; the f64 root rounded once to f32 is the correctly rounded f32 root (numpy parity)
  float n = (float)sqrt((double)__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
  d0 = __fdiv_rn(d0, n); d1 = __fdiv_rn(d1, n); d2 = __fdiv_rn(d2, n);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    dir[r] = __fadd_rn(__fadd_rn(__fmul_rn(R[3 * r], d0), __fmul_rn(R[3 * r + 1], d1)), __fmul_rn(R[3 * r + 2], d2));
  *norm_out = n;
}

__global__ void k_compute_direction(const float* __restrict__ R, const int64_t* __restrict__ idx, int n, int w, double fx,
                                    double fy, double cx, double cy, const float* __restrict__ jitter,
                                    float* __restrict__ dirs, float* __restrict__ norms) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float d[3], nn;
    pixel_direction(R, idx[i], w, fx, fy, cx, cy, jitter != nullptr, jitter ? jitter[2 * i] : 0.f,
                    jitter ? jitter[2 * i + 1] : 0.f, d, &nn);
    dirs[3 * (size_t)i] = d[0]; dirs[3 * (size_t)i + 1] = d[1]; dirs[3 * (size_t)i + 2] = d[2];
    norms[i] = nn;
  }
}

extern "C" int aln_compute_direction(const float* R_WC, const int64_t* idx, int32_t n, int32_t w, double fx, double fy,
                                     double cx, double cy, const float* jitter, float* dirs, float* norms, void* stream) {
  ALN_REQUIRE(R_WC && idx && dirs && norms && w > 0, "compute_direction: bad arguments");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_compute_direction, dim3(aln_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, R_WC, idx, n, w, fx,
                     fy, cx, cy, jitter, dirs, norms);
  ALN_CHECK_LAUNCH("compute_direction");
  return 0;
}

struct RaygenArgs {
  AlnFrames fr; AlnBatch out; int B, chunk, frame_lo, frame_hi;
  uint32_t seed, step; const uint32_t* step_dev;   // step_dev (device memory, may be NULL) is added to step: hipGraph replays
  const int* chunk_frames; const int* ray_idx; const float* jitter;
};
struct RaygenKeys { uint32_t k_frame, k_pix, k_jx, k_jy, k_cls; };

// feature row of (frame, pix): nearest-cell lookup of dataset.py:231-240
__device__ inline const h16* feature_row(const AlnFrames& fr, int frame, int64_t pix) {
  int64_t x = pix % fr.w, y = (pix - x) / fr.w;
  int fx_ = (int)((double)x * ((double)fr.feat_w / (double)fr.w));
  int fy_ = (int)((double)y * ((double)fr.feat_h / (double)fr.h));
  return (const h16*)fr.features + ((size_t)frame * fr.feat_w * fr.feat_h + (size_t)fy_ * fr.feat_w + fx_) * fr.feat_c;
}

template <bool FEATURES = true>
__device__ inline void gather_pixel(const AlnFrames& fr, const AlnBatch& out, int b, int frame, int64_t pix) {
  size_t hw = (size_t)fr.w * fr.h;
  size_t src = (size_t)frame * hw + pix;
  if (out.pixels) {
    out.pixels[3 * (size_t)b] = fr.images[3 * src]; out.pixels[3 * (size_t)b + 1] = fr.images[3 * src + 1];
    out.pixels[3 * (size_t)b + 2] = fr.images[3 * src + 2];
  }
  if (out.depth) out.depth[b] = (float)((double)fr.depths[src] / 1000.0);       // dataset.py:220
  if (out.semantic) out.semantic[b] = (int)fr.semantics[src] - 1;                // dataset.py:221-222
  if constexpr (FEATURES) {
    if (out.features && fr.features) {                                          // dataset.py:231-240
      const h16* f = feature_row(fr, frame, pix);
      for (int c = 0; c < fr.feat_c; ++c) out.features[(size_t)b * fr.feat_c + c] = (float)f[c];
    }
  }
}

// One lane per ray, 64 rays per block (a 4096-ray batch then covers 64 CUs instead of 16: the kernel is a chain of dependent
// random loads per ray and took 21 us however small the batch).  The feature rows -- 64 halves -> 64 floats per ray -- are
// copied by the whole block afterwards, 16 bytes per lane in, 32 bytes out, consecutive lanes on consecutive chunks.
#define RG_BLOCK 64
__global__ __launch_bounds__(RG_BLOCK) void k_raygen_train(RaygenArgs args) {
  __shared__ const h16* frow[RG_BLOCK];
  const uint32_t st = args.step + (args.step_dev ? *args.step_dev : 0u);
  struct A : RaygenArgs, RaygenKeys {} a;
  (RaygenArgs&)a = args;
  a.k_frame = aln_rand_key(args.seed, ALN_STREAM_FRAME, st); a.k_pix = aln_rand_key(args.seed, ALN_STREAM_PIXEL, st);
  a.k_jx = aln_rand_key(args.seed, ALN_STREAM_JX, st); a.k_jy = aln_rand_key(args.seed, ALN_STREAM_JY, st);
  a.k_cls = aln_rand_key(args.seed, ALN_STREAM_CLASS, st);
  const bool feats = a.out.features && a.fr.features;
  for (int b0 = blockIdx.x * RG_BLOCK; b0 < a.B; b0 += gridDim.x * RG_BLOCK) {
    const int b = b0 + threadIdx.x;
    if (b < a.B) {
      int ch = b / a.chunk;
      int frame; int64_t pix;
      // class-weighted chunk (dataset.py:207-211): every ray of the chunk takes the same (class, frame) decision
      bool labelled = !a.chunk_frames && !a.ray_idx && a.fr.n_classes > 0 && a.fr.sem_ratio > 0.f &&
                      aln_rand_uniform(a.k_cls, 3u * (uint32_t)ch) < a.fr.sem_ratio;
      if (labelled) {
        int k = (int)(aln_rand_u32(a.k_cls, 3u * (uint32_t)ch + 1u) % (uint32_t)a.fr.n_classes);
        const int* off = a.fr.cls_offsets + (size_t)k * (a.fr.n_frames + 1);
        int lo_o = off[a.frame_lo], hi_o = off[a.frame_hi];
        if (hi_o > lo_o) {
          int r = lo_o + (int)(aln_rand_u32(a.k_cls, 3u * (uint32_t)ch + 2u) % (uint32_t)(hi_o - lo_o));
          int f0 = a.frame_lo, f1 = a.frame_hi;      // largest f with off[f] <= r  (invariant: off[f0] <= r < off[f1])
          // sixteen-way search: fifteen pivots per step, all requested at once -- two HBM round trips for 200 frames where the binary
          // search made eight, one after the other, and this kernel is nothing but the longest such chain of its 64 waves (round 6)
          while (f1 - f0 > 1) {
            const int stp = (f1 - f0 + 15) / 16;
            int v[15];
#pragma unroll
            for (int j = 0; j < 15; ++j) v[j] = off[min(f0 + stp * (j + 1), f1)];
            int n0 = f0, n1 = f1;
#pragma unroll
            for (int j = 0; j < 15; ++j) { const int m = f0 + stp * (j + 1); if (m < f1 && v[j] <= r) n0 = m; }
#pragma unroll
            for (int j = 14; j >= 0; --j) { const int m = f0 + stp * (j + 1); if (m < f1 && v[j] > r) n1 = m; }
            f0 = n0; f1 = n1;
          }
          frame = f0;
          pix = a.fr.cls_pixels[off[f0] + (int)(aln_rand_u32(a.k_pix, (uint32_t)b) % (uint32_t)(off[f0 + 1] - off[f0]))];
        } else labelled = false;                     // this rank's frame shard has no pixel of that class
      }
      if (!labelled) {
        frame = a.chunk_frames ? a.chunk_frames[ch]
                               : a.frame_lo + (int)(aln_rand_u32(a.k_frame, (uint32_t)ch) % (uint32_t)(a.frame_hi - a.frame_lo));
        pix = a.ray_idx ? a.ray_idx[b] : a.fr.pixel_indices[aln_rand_u32(a.k_pix, (uint32_t)b) % (uint32_t)a.fr.n_pix];
      }
      if (feats) frow[threadIdx.x] = feature_row(a.fr, frame, pix);
      float jx = a.jitter ? a.jitter[2 * b] : aln_rand_uniform(a.k_jx, (uint32_t)b);
      float jy = a.jitter ? a.jitter[2 * b + 1] : aln_rand_uniform(a.k_jy, (uint32_t)b);
      // every read of the ray FIRST, then its stores (round 6): the outputs may alias the frame arrays as far as hipcc knows, so a load
      // written behind a store is issued behind it -- the rotation, the origin, the pixel, its depth and its label were five round trips
      float Rm[9], org[3], rgb[3] = {0.f, 0.f, 0.f};
      const size_t src = (size_t)frame * ((size_t)a.fr.w * a.fr.h) + pix;
#pragma unroll
      for (int k = 0; k < 9; ++k) Rm[k] = a.fr.rotations[9 * (size_t)frame + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) org[k] = a.fr.origins[3 * (size_t)frame + k];
      if (a.out.pixels) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = a.fr.images[3 * src + k];
      }
      uint16_t dep_raw = 0; uint8_t sem_raw = 0;
      if (a.out.depth) dep_raw = a.fr.depths[src];
      if (a.out.semantic) sem_raw = a.fr.semantics[src];
      float d[3], nn;
      pixel_direction(Rm, pix, a.fr.w, a.fr.fx, a.fr.fy, a.fr.cx, a.fr.cy, true, jx, jy, d, &nn);
#pragma unroll
      for (int k = 0; k < 3; ++k) { a.out.rays_d[3 * (size_t)b + k] = d[k]; a.out.rays_o[3 * (size_t)b + k] = org[k]; }
      a.out.norms[b] = nn;
      if (a.out.pixels) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.out.pixels[3 * (size_t)b + k] = rgb[k];
      }
      if (a.out.depth) a.out.depth[b] = (float)((double)dep_raw / 1000.0);       // dataset.py:220
      if (a.out.semantic) a.out.semantic[b] = (int)sem_raw - 1;                  // dataset.py:221-222
    }
    if (feats) {
      __syncthreads();
      const int Cf = a.fr.feat_c, nr = min(RG_BLOCK, a.B - b0);
      float* const dst = a.out.features + (size_t)b0 * Cf;
      if (Cf % 8 == 0 && ((uintptr_t)a.fr.features & 15) == 0 && ((uintptr_t)a.out.features & 15) == 0) {
        const int nch = Cf / 8;
        for (int i = threadIdx.x; i < nr * nch; i += RG_BLOCK) {
          const int r = i / nch, c8 = i % nch;
          const h16x8 v = *(const h16x8*)(frow[r] + 8 * c8);
          float* o = dst + (size_t)r * Cf + 8 * c8;
          *(float4*)o = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
          *(float4*)(o + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        }
      } else {
        for (int i = threadIdx.x; i < nr * Cf; i += RG_BLOCK) dst[i] = (float)frow[i / Cf][i % Cf];
      }
      __syncthreads();
    }
  }
}

__global__ void k_raygen_frame(AlnFrames fr, AlnBatch out, int frame) {
  int hw = fr.w * fr.h;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < hw; b += gridDim.x * blockDim.x) {
    float d[3], nn;
    pixel_direction(fr.rotations + 9 * (size_t)frame, b, fr.w, fr.fx, fr.fy, fr.cx, fr.cy, false, 0.f, 0.f, d, &nn);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      out.rays_d[3 * (size_t)b + k] = d[k];
      out.rays_o[3 * (size_t)b + k] = fr.origins[3 * (size_t)frame + k];
    }
    out.norms[b] = nn;
    AlnBatch o2 = out; o2.features = nullptr;
    gather_pixel(fr, o2, b, frame, b);
  }
}

static int check_frames(const AlnFrames* fr, const AlnBatch* out) {
  ALN_REQUIRE(fr && out, "raygen: NULL descriptor");
  ALN_REQUIRE(fr->images && fr->depths && fr->semantics && fr->rotations && fr->origins, "raygen: NULL frame arrays");
  ALN_REQUIRE(out->rays_o && out->rays_d && out->norms, "raygen: NULL outputs");
  ALN_REQUIRE(fr->n_frames > 0 && fr->w > 0 && fr->h > 0, "raygen: empty frame set");
  return 0;
}

extern "C" int aln_raygen_train(const AlnFrames* fr, const AlnBatch* out, int32_t B, int32_t chunk, int32_t frame_lo,
                                int32_t frame_hi, uint32_t seed, uint32_t step, const int32_t* chunk_frames,
                                const int32_t* ray_idx, const float* jitter, const uint32_t* step_dev, void* stream) {
  if (int rc = check_frames(fr, out)) return rc;
  ALN_REQUIRE(chunk > 0 && B % chunk == 0, "raygen_train: batch %d is not a multiple of chunk %d", B, chunk);
  ALN_REQUIRE(chunk_frames || (0 <= frame_lo && frame_lo < frame_hi && frame_hi <= fr->n_frames), "raygen_train: bad frame range");
  ALN_REQUIRE(ray_idx || (fr->pixel_indices && fr->n_pix > 0), "raygen_train: pixel_indices missing");
  ALN_REQUIRE(fr->n_classes == 0 || (fr->cls_offsets && fr->cls_pixels), "raygen_train: class index arrays missing");
  if (B <= 0) return 0;
  RaygenArgs a{*fr, *out, B, chunk, frame_lo, frame_hi, seed, step, step_dev, chunk_frames, ray_idx, jitter};
  hipLaunchKernelGGL(k_raygen_train, dim3(aln_grid_for(B, RG_BLOCK)), dim3(RG_BLOCK), 0, (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("raygen_train");
  return 0;
}

extern "C" int aln_raygen_frame(const AlnFrames* fr, const AlnBatch* out, int32_t frame, void* stream) {
  if (int rc = check_frames(fr, out)) return rc;
  ALN_REQUIRE(frame >= 0 && frame < fr->n_frames, "raygen_frame: frame %d out of range", frame);
  hipLaunchKernelGGL(k_raygen_frame, dim3(aln_grid_for((int64_t)fr->w * fr->h, 256)), dim3(256), 0, (hipStream_t)stream, *fr,
                     *out, frame);
  ALN_CHECK_LAUNCH("raygen_frame");
  return 0;
}
