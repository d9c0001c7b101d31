// Bias-free ReLU MLPs on MFMA (v_mfma_f32_32x32x16_f16), forward + backward.
// Replaces tcnn Network{FullyFusedMLP,CutlassMLP} instantiated at autolabel/models.py:84-136
// (sigma_net 48->128->128->16, color_net 32->128->128->16, semantic_features 16->D->D->D,
//  semantic_out (D+16)->64->Cpad).  Spec: oracle/nerf_oracle.py:mlp_forward (fp16 weights and
// activations, fp32 accumulate, input padded with ones by the producer of x).
//
// Formulation: Out^T[n, s] = W[n, k] * Act^T[k, s].  A = weights (32 output features x 16 k),
// B = activations (16 k x 32 samples): lane (s = l&31, hf = l>>5) owns sample s for the whole chain.
// The C layout of one layer (lane owns features 32mb + 8q + 4hf + r of its sample) is consumed
// directly as the B operand of the next layer -- the k index of an MFMA is only a summation index, so
// the weights are pre-permuted ("chained kmap") instead of transposing activations through LDS.
// Activations therefore never leave registers between layers; weights live in LDS in fragment order
// (one conflict-free ds_read_b128 per lane per MFMA).
#include "mlp_shared.h"

extern "C" int64_t aln_mlp_frag_halves(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden,
                                       int32_t backward) {
  MlpLayers L = mlp_layers(in_pad, hidden, out_pad, n_hidden);
  size_t frags = backward ? bwd_frag_off(L, -1) : fwd_frag_off(L, L.n);
  return (int64_t)frags * 512;
}

// row-major fp16 copy of the weights: read straight for forward fragments and through ds_read_b64_tr_b16 for the transposed
// (backward) fragments -- one LDS image serves both directions.  W_0 is [out][in + 8] (pitch padded by 8 halves); the
// matrices whose input is the hidden layer are [out][hid_pitch(hid)].
__host__ __device__ inline int wrow_pitch(const MlpLayers& L, int l) { return l == 0 ? L.in_[0] + 8 : hid_pitch(L.in_[l]); }
__host__ __device__ inline size_t wrow_off(const MlpLayers& L, int l) {
  size_t o = 0; for (int i = 0; i < l; ++i) o += (size_t)L.out_[i] * wrow_pitch(L, i); return o;
}
// element i of the row-major image / element e of the fragment images of one head
__device__ inline h16 rowmajor_elem(const float* __restrict__ w, const MlpLayers& L, size_t i) {
  int l = 0; while (l + 1 < L.n && i >= wrow_off(L, l + 1)) ++l;
  size_t e = i - wrow_off(L, l); int pitch = wrow_pitch(L, l); int o = (int)(e / pitch), k = (int)(e % pitch);
  return (h16)(k < L.in_[l] ? w[L.w_off[l] + (size_t)o * L.in_[l] + k] : 0.f);
}
__device__ inline h16 frag_elem(const float* __restrict__ w, const MlpLayers& L, size_t e, bool bw) {
  int j = e & 7, lane = (e >> 3) & 63; size_t frag = e >> 9;
  int hf = lane >> 5, c = lane & 31;
  int l = 0; size_t base = 0;
  if (!bw) {
    for (l = 0; l < L.n; ++l) { size_t cnt = (size_t)ceil32(L.out_[l]) * (L.in_[l] / 16); if (frag < base + cnt) break; base += cnt; }
    int KS = L.in_[l] / 16; int mb = (frag - base) / KS, ks = (frag - base) % KS;
    int o = 32 * mb + c, k = (l == 0) ? kmap_natural(ks, hf, j) : kmap_chained(ks, hf, j);
    return (h16)((o < L.out_[l] && k < L.in_[l]) ? w[L.w_off[l] + (size_t)o * L.in_[l] + k] : 0.f);
  }
  for (l = L.n - 1; l >= 0; --l) { size_t cnt = (size_t)ceil32(L.in_[l]) * (L.out_[l] / 16); if (frag < base + cnt) break; base += cnt; }
  int KS = L.out_[l] / 16; int mb = (frag - base) / KS, ks = (frag - base) % KS;
  int ii = 32 * mb + c, o = (l == L.n - 1) ? kmap_natural(ks, hf, j) : kmap_chained(ks, hf, j);
  return (h16)((ii < L.in_[l] && o < L.out_[l]) ? w[L.w_off[l] + (size_t)o * L.in_[l] + ii] : 0.f);
}

__global__ void k_mlp_rowmajor(const float* __restrict__ w, MlpLayers L, h16* __restrict__ wr) {
  size_t total = wrow_off(L, L.n);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) wr[i] = rowmajor_elem(w, L, i);
}

__global__ void k_mlp_repack(const float* __restrict__ w, MlpLayers L, h16* __restrict__ wf, h16* __restrict__ wb) {
  // one thread per (frag, lane, j)
  size_t nf = fwd_frag_off(L, L.n) * 512, nb = bwd_frag_off(L, -1) * 512;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nf + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nf) wf[i] = frag_elem(w, L, i, false); else wb[i - nf] = frag_elem(w, L, i - nf, true);
  }
}

// all heads of a model in ONE launch (the per-step shadow refresh after the optimizer: 8 launches -> 1)
#define ALN_MAX_HEADS 8
struct RepackHead { const float* w; MlpLayers L; h16* wf; h16* wb; h16* wr; size_t nf, nb, nr; };
struct RepackAll { int n; RepackHead h[ALN_MAX_HEADS]; };
__global__ void k_mlp_repack_all(RepackAll a) {
  size_t total = 0;
  for (int k = 0; k < a.n; ++k) total += a.h[k].nf + a.h[k].nb + a.h[k].nr;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t e = i; int k = 0;
    while (e >= a.h[k].nf + a.h[k].nb + a.h[k].nr) { e -= a.h[k].nf + a.h[k].nb + a.h[k].nr; ++k; }
    const RepackHead& h = a.h[k];
    if (e < h.nf) h.wf[e] = frag_elem(h.w, h.L, e, false);
    else if (e < h.nf + h.nb) h.wb[e - h.nf] = frag_elem(h.w, h.L, e - h.nf, true);
    else h.wr[e - h.nf - h.nb] = rowmajor_elem(h.w, h.L, e - h.nf - h.nb);
  }
}

extern "C" int64_t aln_mlp_rowmajor_halves(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden) {
  MlpLayers L = mlp_layers(in_pad, hidden, out_pad, n_hidden);
  return (int64_t)wrow_off(L, L.n);
}

extern "C" int aln_mlp_repack(const float* w_master, int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden,
                              void* wf, void* wb, void* wr, void* stream) {
  ALN_REQUIRE(w_master && wf && wb, "mlp_repack: NULL pointer");
  ALN_REQUIRE(in_pad % 16 == 0 && out_pad % 16 == 0 && (hidden == 64 || hidden == 128) && (n_hidden == 1 || n_hidden == 2),
              "mlp_repack: unsupported shape in=%d hid=%d out=%d nh=%d", in_pad, hidden, out_pad, n_hidden);
  MlpLayers L = mlp_layers(in_pad, hidden, out_pad, n_hidden);
  hipLaunchKernelGGL(k_mlp_repack, dim3(64), dim3(256), 0, (hipStream_t)stream, w_master, L, (h16*)wf, (h16*)wb);
  ALN_CHECK_LAUNCH("mlp_repack");
  if (wr) {
    hipLaunchKernelGGL(k_mlp_rowmajor, dim3(32), dim3(256), 0, (hipStream_t)stream, w_master, L, (h16*)wr);
    ALN_CHECK_LAUNCH("mlp_rowmajor");
  }
  return 0;
}

extern "C" int aln_mlp_repack_all(int32_t n_heads, const float* const* w_master, const AlnMlpDesc* const* descs, void* stream) {
  ALN_REQUIRE(n_heads >= 0 && n_heads <= ALN_MAX_HEADS && (n_heads == 0 || (w_master && descs)), "mlp_repack_all: bad arguments");
  if (n_heads == 0) return 0;
  RepackAll a; a.n = n_heads;
  size_t total = 0;
  for (int k = 0; k < n_heads; ++k) {
    const AlnMlpDesc* m = descs[k];
    ALN_REQUIRE(m && w_master[k] && m->wf && m->wb, "mlp_repack_all: NULL pointer in head %d", k);
    ALN_REQUIRE(m->in_pad % 16 == 0 && m->out_pad % 16 == 0 && (m->hidden == 64 || m->hidden == 128) &&
                    (m->n_hidden == 1 || m->n_hidden == 2), "mlp_repack_all: unsupported shape in head %d", k);
    RepackHead& h = a.h[k];
    h.w = w_master[k]; h.L = mlp_layers(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
    h.wf = (h16*)m->wf; h.wb = (h16*)m->wb; h.wr = (h16*)m->wr;
    h.nf = fwd_frag_off(h.L, h.L.n) * 512; h.nb = bwd_frag_off(h.L, -1) * 512; h.nr = m->wr ? wrow_off(h.L, h.L.n) : 0;
    total += h.nf + h.nb + h.nr;
  }
  int g = (int)((total + 255) / 256); if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_mlp_repack_all, dim3(g), dim3(256), 0, (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("mlp_repack_all");
  return 0;
}

// ---------------------------------------------------------------- device helpers
__device__ inline void copy_to_lds(h16* lds, const h16* g, size_t halves) {
  const uint4* s = (const uint4*)g; uint4* d = (uint4*)lds;
  for (size_t i = threadIdx.x; i < halves / 8; i += blockDim.x) d[i] = s[i];
}
template <int NB>
__device__ inline void zero_acc(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}
template <int NB>
__device__ inline void relu_pack_store(f32x16 (&acc)[NB], h16x8 (&p)[2 * NB], h16* dst_row /*row base or null*/, int hf) {
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) ((uint32_t*)&p[2 * m + (r >> 3)])[(r & 7) >> 1] = relu2(acc[m][r], acc[m][r + 1]);
    if (dst_row) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t* w = (const uint32_t*)&p[2 * m + (q >> 1)];
        *(u32x2*)(dst_row + 32 * m + 8 * q + 4 * hf) = (u32x2){w[2 * (q & 1)], w[2 * (q & 1) + 1]};
      }
    }
  }
}
// ---------------------------------------------------------------- virtual input rows
// The heads' inputs / output gradients are cheap functions of tensors that already exist (models.py:248-256 plumbing);
// instead of materialising them ([rows,80] + [rows,64] + [rows,16] fp16 round trips through HBM) the MLP kernels build
// each 8-feature chunk on the fly.
enum { SRC_PLAIN = 0, SRC_SEMF_IN = 1, SRC_SEMO_IN = 2, SRC_DLOGITS = 3, SRC_DSEMF_OUT = 4, SRC_COLOR_IN = 5 };
// n / d for a divisor fixed at launch (Granlund-Montgomery, round-up method): one multiply-high and two shifts instead of the
// ~35-instruction software division -- the row sources below turn a sample row into its ray for every 16-byte chunk they build
struct FastDiv { uint32_t m, s1, s2; };
static FastDiv fastdiv_make(uint32_t d) {
  if (d == 0) d = 1;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1; f.s2 = l - f.s1;
  return f;
}
__device__ inline uint32_t fastdiv(uint32_t n, FastDiv f) {
  const uint32_t t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
struct RowSrc {
  int mode;
  const h16* a; int lda;      // PLAIN: x ; SEMF_IN: sigma_out (ld 16) ; SEMO_IN: f (ld D)
  const h16* b; int ldb;      // SEMO_IN: sigma_out ; DSEMF_OUT: d_semo_in (its first D columns = dL/df, ReLU mask applied)
  const float* w_row; const float* g;   // DLOGITS / DSEMF_OUT: per-row weight, per-ray output gradient [N, gw]
  int N, S1, S2, D, G, gw;
  FastDiv d1, d2;             // / S1, / S2 (set by row_src_rays)
  const int* idx;             // COLOR_IN: live_idx (compact row -> sample row) or NULL; b = sigma_out, g = directions
  int fold_geo;               // DSEMF_OUT (as dL/dout of semantic_features): the backward adds b[row][D .. D+16) -- the geo_feat
};                            //           columns of d(semantic_out input) -- into its d_in rows (one d(geo_feat) tensor leaves)                            //           (gw = 0: one per ray [N,3], gw = 1: one per sample row [rows,3])
static void row_src_rays(RowSrc& s, int N, int S1, int S2) {
  s.N = N; s.S1 = S1; s.S2 = S2 > 0 ? S2 : 1; s.d1 = fastdiv_make((uint32_t)s.S1); s.d2 = fastdiv_make((uint32_t)s.S2);
}
__device__ inline int row_ray(const RowSrc& s, int row) {   // pass-major rows: N x S1, then N x S2
  const int n1 = s.N * s.S1;
  return row < n1 ? (int)fastdiv((uint32_t)row, s.d1) : (int)fastdiv((uint32_t)(row - n1), s.d2);
}
// per-ray gradient row g[ray][c0 .. c0 + 7] (fp32): two 16-byte loads when the row width keeps every chunk 32-byte aligned (the
// 64-wide feature gradient: eight chunks per sample row), else element by element with the tail masked (7 class logits)
__device__ inline void ray_grad8(const RowSrc& s, int ray, int c0, float* o) {
  const float* g = s.g + (size_t)ray * s.gw + c0;
  if ((s.gw & 7) == 0) {
    const float4 a = *(const float4*)g, b = *(const float4*)(g + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (c0 + j < s.gw) ? g[j] : 0.f;
  }
}
__device__ inline h16x8 geo_chunk(const h16* sigma_out, size_t row, int j0, int G) {
  // [geo_feat (G), 1, 1, ...] features j0..j0+7 ; geo_feat[g] = sigma_out[row][1 + g].  j0 is 0 or 8: both cases use
  // compile-time element indices (a runtime-indexed local array would live in scratch memory)
  const h16* r = sigma_out + row * 16;
  const h16x8 lo = *(const h16x8*)r, hi = *(const h16x8*)(r + 8);
  h16x8 o;
  if (j0 == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { h16 v = (j < 7) ? lo[j + 1] : hi[0]; o[j] = (j < G) ? v : (h16)1.0f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { h16 v = (j < 7) ? hi[j + 1] : (h16)1.0f; o[j] = (8 + j < G) ? v : (h16)1.0f; }
  }
  return o;
}
__device__ inline h16x8 load_chunk8(const RowSrc& s, int row, int c0) {
  switch (s.mode) {
    case SRC_SEMF_IN: return geo_chunk(s.a, (size_t)row, c0, s.G);
    case SRC_SEMO_IN: {
      if (c0 >= s.D) return geo_chunk(s.b, (size_t)row, c0 - s.D, s.G);
      h16x8 v = *(const h16x8*)(s.a + (size_t)row * s.lda + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)v[j] > 0.f ? v[j] : (h16)0.f;
      return v;
    }
    case SRC_DLOGITS: case SRC_DSEMF_OUT: {
      const float w = s.w_row[row];
      float g[8];
      ray_grad8(s, row_ray(s, row), c0, g);
      h16x8 o;
      if (s.mode == SRC_DLOGITS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)((c0 + j < s.gw) ? w * g[j] : 0.f);
      } else {
        h16x8 d = *(const h16x8*)(s.b + (size_t)row * s.ldb + c0);   // d(semantic_out) / d f, ReLU mask applied by its producer
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)(w * g[j] + (float)d[j]);
      }
      return o;
    }
    case SRC_COLOR_IN: {   // color_net input [SH16(dir), geo_feat, 1...] of models.py:205-212, built on the fly (inference)
      const int r = s.idx ? s.idx[row] : row;
      if (c0 >= 16) return geo_chunk(s.b, (size_t)r, c0 - 16, s.G);
      const float* d = s.g + 3 * (size_t)(s.gw ? r : row_ray(s, r));
      float sh[16];
      sh4_of_dir(d, sh);
      h16x8 o;
      if (c0 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)sh[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)sh[8 + j];
      }
      return o;
    }
    default: return *(const h16x8*)(s.a + (size_t)row * s.lda + c0);
  }
}
static RowSrc plain_src(const void* p, int ld) {
  RowSrc s{}; s.mode = SRC_PLAIN; s.a = (const h16*)p; s.lda = ld; return s;
}
__device__ inline void load_tile_src(lds_h16* tile, int pitch, const RowSrc& s, int ncols, int r0, int nrows, int limit) {
  const int per_row = ncols / 8;
  for (int i = threadIdx.x; i < nrows * per_row; i += blockDim.x) {
    int r = i / per_row, k = i % per_row;
    h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + r < limit) v = load_chunk8(s, r0 + r, 8 * k);
    *(LDS_VEC(h16x8)*)(tile + r * pitch + 8 * k) = v;
  }
}

// Two-stage form of load_chunk8 for software prefetch: raw_load only issues the global loads of one 8-feature chunk (no
// dependent arithmetic, so nothing waits on them), raw_finish turns the registers into the fp16 chunk when the tile is
// stashed into LDS one tile later.  MODE is a compile-time copy of RowSrc::mode so the unused fields fold away.
struct RawChunk { h16x8 a, b; float w; float g[8]; int ray; };
// LATEG: only the streamed operands (a, b, w) are requested ahead; the per-ray gradient row (L1 / L2 resident: a ray spans S
// consecutive rows) is read in raw_finish -- 9 fewer registers per chunk held across the tile.
template <int MODE, bool LATEG = false>
__device__ inline void raw_load(RawChunk& r, const RowSrc& s, int row, int c0) {
  if constexpr (MODE == SRC_PLAIN) {
    r.a = *(const h16x8*)(s.a + (size_t)row * s.lda + c0);
  } else if constexpr (MODE == SRC_SEMF_IN) {
    const h16* p = s.a + (size_t)row * 16; r.a = *(const h16x8*)p; r.b = *(const h16x8*)(p + 8);
  } else if constexpr (MODE == SRC_SEMO_IN) {
    if (c0 >= s.D) { const h16* p = s.b + (size_t)row * 16; r.a = *(const h16x8*)p; r.b = *(const h16x8*)(p + 8); }
    else r.a = *(const h16x8*)(s.a + (size_t)row * s.lda + c0);
  } else {
    const int ray = row_ray(s, row);
    r.w = s.w_row[row];
    if constexpr (LATEG) r.ray = ray;
    else ray_grad8(s, ray, c0, r.g);
    if constexpr (MODE == SRC_DSEMF_OUT) r.b = *(const h16x8*)(s.b + (size_t)row * s.ldb + c0);
  }
}
__device__ inline h16x8 geo_from(const h16x8& lo, const h16x8& hi, int j0, int G) {   // geo_chunk on loaded registers
  h16x8 o;
  if (j0 == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { h16 v = (j < 7) ? lo[j + 1] : hi[0]; o[j] = (j < G) ? v : (h16)1.0f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { h16 v = (j < 7) ? hi[j + 1] : (h16)1.0f; o[j] = (8 + j < G) ? v : (h16)1.0f; }
  }
  return o;
}
#define GT_RAYS 2   // per-ray gradient rows staged in LDS per tile (a 128-row tile spans <= 2 rays when a ray has >= 128 samples)
template <int MODE, bool LATEG = false>
__device__ inline h16x8 raw_finish(const RawChunk& r, const RowSrc& s, int c0, const float* gt = nullptr, int gt_first = 0, int gt_w = 0) {
  if constexpr (MODE == SRC_PLAIN) return r.a;
  else if constexpr (MODE == SRC_SEMF_IN) return geo_from(r.a, r.b, c0, s.G);
  else if constexpr (MODE == SRC_SEMO_IN) {
    if (c0 >= s.D) return geo_from(r.a, r.b, c0 - s.D, s.G);
    h16x8 v = r.a;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)v[j] > 0.f ? v[j] : (h16)0.f;
    return v;
  } else {
    h16x8 o;
    float gl[8];
    if constexpr (LATEG) {
      const unsigned dr = (unsigned)(r.ray - gt_first);
      if (gt && dr < GT_RAYS) {   // the tile's rays: staged in LDS by the kernel (k_mlp_bwd_recomp8)
        const float4 a = *(const float4*)(gt + dr * gt_w + c0), b = *(const float4*)(gt + dr * gt_w + c0 + 4);
        gl[0] = a.x; gl[1] = a.y; gl[2] = a.z; gl[3] = a.w; gl[4] = b.x; gl[5] = b.y; gl[6] = b.z; gl[7] = b.w;
      } else ray_grad8(s, r.ray, c0, gl);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) gl[j] = r.g[j];
    }
    if constexpr (MODE == SRC_DLOGITS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (h16)((c0 + j < s.gw) ? r.w * gl[j] : 0.f);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (h16)(r.w * gl[j] + (float)r.b[j]);
    }
    return o;
  }
}

// One layer of the register chain with software-pipelined A fragments: the NB fragments of k-step ks+1 are requested
// before the NB MFMAs of k-step ks are issued (one wave per SIMD: nothing else hides the ~150-cycle LDS latency).
template <int NB, int KSN, class FragFn, class BFn>
__device__ inline void chain_layer(f32x16 (&acc)[NB], FragFn frag, BFn bop) {
  h16x8 a[2][NB], b[2];
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < NB; ++m) a[0][m] = frag(m, 0);
  b[0] = bop(0);
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks) {
    if (ks + 1 < KSN) {   // next k-step's A fragments AND B operand (an LDS tile read for the first / last layer) in flight
#pragma unroll
      for (int m = 0; m < NB; ++m) a[(ks + 1) & 1][m] = frag(m, ks + 1);
      b[(ks + 1) & 1] = bop(ks + 1);
    }
    // pin the order: left alone, the machine scheduler turns this into NB dependent accumulator chains with one
    // "ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma" round trip per MFMA (~80 cycles each instead of 32)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NB; ++m) acc[m] = mfma16(a[ks & 1][m], b[ks & 1], ks == 0 ? zero : acc[m]);   // C = 0 inline: no accumulator zeroing
    __builtin_amdgcn_sched_barrier(0);
  }
}
// The same with RPW 32-row sub-tiles per wave: every weight fragment read from LDS feeds RPW MFMAs (independent accumulators,
// independent B operands), so a phase carries RPW times the work behind the same dependent-latency chain.
template <int RPW, int NB, int KSN, class FragFn, class BFn>
__device__ inline void chain_layer_m(f32x16 (&acc)[RPW][NB], FragFn frag, BFn bop) {
  h16x8 a[2][NB], b[2][RPW];
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < NB; ++m) a[0][m] = frag(m, 0);
#pragma unroll
  for (int u = 0; u < RPW; ++u) b[0][u] = bop(u, 0);
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks) {
    if (ks + 1 < KSN) {
#pragma unroll
      for (int m = 0; m < NB; ++m) a[(ks + 1) & 1][m] = frag(m, ks + 1);
#pragma unroll
      for (int u = 0; u < RPW; ++u) b[(ks + 1) & 1][u] = bop(u, ks + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int u = 0; u < RPW; ++u) acc[u][m] = mfma16(a[ks & 1][m], b[ks & 1][u], ks == 0 ? zero : acc[u][m]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// ---------------------------------------------------------------- forward
// (three 256-thread blocks per CU wherever the chain fits 168 VGPRs -- the 128-wide heads with up to 48 inputs: the register
//  allocator is told so, one register over costs a third of the resident waves)
// PLAIN: the input rows are a plain [rows, in_pad] fp16 matrix (every training-step launch): one 16-byte load per chunk instead of
// the row-source switch (210 branches and 1 170 scalar instructions in the 48-wide instantiation: three blocks per CU share one
// scalar unit)
template <int HID, int NHID, int KS0, bool PLAIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((HID == 128 && NHID == 2 && KS0 > 3) ? 2 : 3))) void k_mlp_fwd(const h16* __restrict__ wf_g, size_t wf_halves, int in_pad, int out_pad,
                                                RowSrc xs, int rows, const int* __restrict__ rows_dev,
                                                h16* __restrict__ h1, h16* __restrict__ h2, h16* __restrict__ out,
                                                float* __restrict__ sigma) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* wl = (h16*)smem;
  copy_to_lds(wl, wf_g, wf_halves);
  __syncthreads();
  constexpr int NB = HID / 32, KS = HID / 16;
  const h16x8* frag0 = (const h16x8*)wl;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, c = lane & 31;
  if (rows_dev) rows = min(rows, *rows_dev);
  const int OB = ceil32(out_pad);
  const size_t f1 = (size_t)NB * KS0;                       // layer-1 frags start
  const size_t fl = f1 + (NHID == 2 ? (size_t)NB * KS : 0); // last-layer frags start
  const int ntiles = (rows + 31) / 32;
  // The input chunks of a tile are all requested together, and the NEXT tile's chunks are requested into the same
  // registers as soon as layer 0 has consumed them: their latency hides behind the remaining layers of this tile.
  h16x8 xb[KS0];
  auto load_x = [&](int t) {
    const int r = t * 32 + c;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      if constexpr (PLAIN) xb[ks] = (r < rows) ? *(const h16x8*)(xs.a + (size_t)r * xs.lda + 16 * ks + 8 * hf) : z;
      else xb[ks] = (r < rows) ? load_chunk8(xs, r, 16 * ks + 8 * hf) : z;
    }
  };
  const int tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  // The colour head's rows built on the fly (SRC_COLOR_IN, the render path), STAGED (round 6): load_chunk8 resolves a chunk on the spot --
  // compact index -> sample row -> ray direction + sigma_out row -> SH-4 and the geo_feat shift, two dependent round trips and the
  // arithmetic inside what was meant as a prefetch.  Here the sample row of tile t + 2 and the raw direction / sigma_out row of tile
  // t + 1 are in flight while tile t is in its chain; every load is unconditional (clamped tile and row).
  const bool staged = !PLAIN && KS0 == 2 && xs.mode == SRC_COLOR_IN && xs.idx != nullptr;   // (with the compaction's row list: a `idx ? idx[r] : r` would be a load in a branch again)
  int ridx_n = 0; float dn[3] = {0.f, 0.f, 0.f}; h16x8 lon = {0, 0, 0, 0, 0, 0, 0, 0}, hin = lon;
  auto stage_idx = [&](int t) { return xs.idx[min(min(t, ntiles - 1) * 32 + c, rows - 1)]; };
  auto stage_raw = [&](int ridx) {
    const float* d = xs.g + 3 * (size_t)(xs.gw ? ridx : row_ray(xs, ridx));
    dn[0] = d[0]; dn[1] = d[1]; dn[2] = d[2];
    lon = *(const h16x8*)(xs.b + (size_t)ridx * 16); hin = *(const h16x8*)(xs.b + (size_t)ridx * 16 + 8);
  };
  auto stage_finish = [&](bool ok) {   // chunk 0: SH-4 values 8 hf .. 8 hf + 7; chunk 1: [geo_feat, 1...] values 8 hf .. 8 hf + 7
    float sh[16];
    sh4_of_dir(dn, sh);
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    h16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = (h16)(hf ? sh[8 + j] : sh[j]);
      const h16 v0 = (j < 7) ? lon[j + 1] : hin[0], v1 = (j < 7) ? hin[j + 1] : (h16)1.0f;
      o1[j] = (8 * hf + j < xs.G) ? (hf ? v1 : v0) : (h16)1.0f;
    }
    xb[0] = ok ? o0 : z; xb[KS0 - 1] = ok ? o1 : z;
  };
  if (tile < ntiles) {
    if (staged) { stage_raw(stage_idx(tile)); ridx_n = stage_idx(tile + tstride); }
    else load_x(tile);
  }
  for (; tile < ntiles; tile += tstride) {
    const int row = tile * 32 + c;
    const bool valid = row < rows;
    // the weight fragments are re-read from LDS every tile: an offset the optimiser cannot see through keeps it from
    // hoisting all of them into registers (306 VGPRs -> one wave per SIMD, nothing to overlap the pack/store phases with)
    int fo = 0;
    asm volatile("" : "+v"(fo));
    const h16x8* frag = frag0 + fo;
    f32x16 acc[NB];
    if (staged) {
      stage_finish(valid);
      stage_raw(ridx_n);                          // tile t + 1: its sample row arrived a trip ago
      ridx_n = stage_idx(tile + 2 * tstride);     // tile t + 2
    }
    chain_layer<NB, KS0>(acc, [&](int m, int ks) { return frag[((size_t)m * KS0 + ks) * 64 + lane]; }, [&](int ks) { return xb[ks]; });
    if (!staged && tile + tstride < ntiles) load_x(tile + tstride);
    h16x8 p[KS];
    relu_pack_store<NB>(acc, p, (valid && h1) ? h1 + (size_t)row * HID : nullptr, hf);
    if (NHID == 2) {
      chain_layer<NB, KS>(acc, [&](int m, int ks) { return frag[(f1 + (size_t)m * KS + ks) * 64 + lane]; }, [&](int ks) { return p[ks]; });
      relu_pack_store<NB>(acc, p, (valid && h2) ? h2 + (size_t)row * HID : nullptr, hf);
    }
    for (int ob = 0; ob < OB; ++ob) {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) o = mfma16(frag[(fl + (size_t)ob * KS + ks) * 64 + lane], p[ks], o);
      if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int f = 32 * ob + 8 * q + 4 * hf;
          if (f < out_pad) {
            h16x4 v; v[0] = (h16)o[4 * q]; v[1] = (h16)o[4 * q + 1]; v[2] = (h16)o[4 * q + 2]; v[3] = (h16)o[4 * q + 3];
            *(h16x4*)(out + (size_t)row * out_pad + f) = v;
            // density head (models.py:175-188): sigma = trunc_exp(h0) of the fp16 output, written by the lane that holds feature 0
            if (sigma && f == 0) sigma[row] = expf((float)v[0]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- both semantic heads, forward, one kernel
// semantic_features (G -> 64 -> 64 -> D = 64) and semantic_out (cat[relu(f), geo_feat] -> 64 -> C) of models.py:248-256 on the
// same 32-row tile: f goes through a wave-private LDS tile, from which it is (a) written out as whole 128-byte rows and
// (b) read back -- ReLU applied -- in the natural k-order of semantic_out's first layer, so both heads keep their ordinary
// fragment images and f is never re-read from HBM (the two-launch path reads it back: 128 B/sample, 24 % of a render pass).
// SUMS (the training step): neither f nor the logits leave the CU.  A tile's 32 rows belong to one ray (both sample counts are
// multiples of 32), and all the step needs of them is the ray's weighted sums sum_s w_s f_s, sum_s w_s logits_s (models.py:
// 195-203): the wave leaves the partial sums of its tile -- [1 x 32] x [32 x 64] through the matrix pipe, the weights as row 0
// (fp16 head) and row 1 (fp16 remainder of the fp32 weight) of the A operand, the tile read with the transpose load; the logits,
// whose rows are lanes, through a lane butterfly -- in tile_sums[tile][96] (64 features, <= 32 logits), 384 B per tile instead of
// 5 KB of rows; aln_composite_out adds a ray's tiles
// in a fixed order.  The backward recomputes f and the logits anyway (k_sem_bwd_pair, which also hands <f, g_feat> + <logits,
// g_sem> per row to the compositing backward).
template <int KSG, bool SUMS>   // KSG: k-steps of the geo_feat input of both heads (in_pad of semantic_features / 16 = 1)
__global__ __launch_bounds__(256) void k_sem_fwd_fused(const h16* __restrict__ wf_f, size_t halves_f, const h16* __restrict__ wf_o,
                                                      size_t halves_o, const h16* __restrict__ sigma_out, int rows, int G,
                                                      int out_pad_o, h16* __restrict__ feat, h16* __restrict__ logits,
                                                      const float* __restrict__ w_row, float* __restrict__ tile_sums) {
  constexpr int HID = 64, D = 64, NB = 2, KS = 4, PT = D + 8;   // PT: pitch of the f tile (row-per-lane 16-byte reads conflict-free)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* wl_f = (h16*)smem;
  h16* wl_o = wl_f + ((halves_f + 7) & ~(size_t)7);
  lds_h16* ftile = (lds_h16*)(wl_o + ((halves_o + 7) & ~(size_t)7)) + (threadIdx.x >> 6) * (32 * PT);
  copy_to_lds(wl_f, wf_f, halves_f);
  copy_to_lds(wl_o, wf_o, halves_o);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, c = lane & 31;
  const int OBO = ceil32(out_pad_o);
  constexpr int KS0O = D / 16 + KSG;                        // semantic_out layer 0: relu(f) k-steps + the geo_feat k-step
  const size_t f1 = (size_t)NB * KSG, fl = f1 + (size_t)NB * KS;   // semantic_features: layer 1 / last layer fragments
  const size_t flo = (size_t)NB * KS0O;                             // semantic_out: last layer fragments
  const int ntiles = (rows + 31) / 32, tstride = gridDim.x * 4;
  static_assert(KSG == 1, "one geo_feat k-step");
  // The next tile's inputs are REQUESTED while this tile is in its chain and not touched before the next trip (round 6): the raw
  // sigma_out row (geo_chunk() shuffled it at once -- a use right behind the load is a wait right behind it, the "prefetch" waited out its
  // own round trip), and for SUMS the tile's weights, which were loaded and converted at the top of every tile.  Unconditional loads
  // from clamped rows / tiles (a load in a branch makes hipcc wait with vmcnt(0) at the join).
  h16x8 xlo, xhi; float wln = 0.f; float4 wq[4];
  auto load_x = [&](int t) {
    t = min(t, ntiles - 1);
    const int r = min(t * 32 + c, rows - 1);
    xlo = *(const h16x8*)(sigma_out + (size_t)r * 16); xhi = *(const h16x8*)(sigma_out + (size_t)r * 16 + 8);
    if constexpr (SUMS) {   // (rows is a multiple of 32 here: whole tiles)
      wln = w_row[r];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { wq[2 * ks] = *(const float4*)(w_row + t * 32 + 16 * ks + 8 * hf); wq[2 * ks + 1] = *(const float4*)(w_row + t * 32 + 16 * ks + 8 * hf + 4); }
    }
  };
  auto geo_of = [&](h16x8 lo, h16x8 hi) __attribute__((always_inline)) {   // geo_chunk() of the raw row for this lane's half, without a divergent branch
    h16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const h16 v0 = (j < 7) ? lo[j + 1] : hi[0], v1 = (j < 7) ? hi[j + 1] : (h16)1.0f;
      const h16 v = hf ? v1 : v0;
      o[j] = (8 * hf + j < G) ? v : (h16)1.0f;
    }
    return o;
  };
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) load_x(tile);
  const PlainV<lds_h16*> ftv{ftile, PT};
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (; tile < ntiles; tile += tstride) {
    const int row0 = tile * 32, row = row0 + c;
    const bool valid = row < rows;
    int fo = 0;
    asm volatile("" : "+v"(fo));                             // keep the fragments in LDS (see k_mlp_fwd)
    const h16x8* ff = (const h16x8*)wl_f + fo;
    const h16x8* fo_ = (const h16x8*)wl_o + fo;
    f32x16 acc[NB];
    h16x8 p[KS], geo[KSG];
    h16x8 aw[2];   // SUMS: A operand of the weighted row sum, k-step ks = rows 16 ks .. 16 ks + 15: row 0 = fp16(w), row 1 = fp16(w - row 0)
    float wl = 0.f;   // SUMS: the weight of this lane's row
    if constexpr (SUMS) {
      wl = wln;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 w0 = wq[2 * ks], w1 = wq[2 * ks + 1];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { const h16 hi = (h16)wv[j]; aw[ks][j] = c == 0 ? hi : (c == 1 ? (h16)(wv[j] - (float)hi) : (h16)0.f); }
      }
    }
    {
      const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      geo[0] = valid ? geo_of(xlo, xhi) : z;
    }
    load_x(tile + tstride);   // (clamped to the last tile when there is no next one)
    chain_layer<NB, KSG>(acc, [&](int m, int ks) { return ff[((size_t)m * KSG + ks) * 64 + lane]; }, [&](int ks) { return geo[ks]; });
    relu_pack_store<NB>(acc, p, nullptr, hf);
    chain_layer<NB, KS>(acc, [&](int m, int ks) { return ff[(f1 + (size_t)m * KS + ks) * 64 + lane]; }, [&](int ks) { return p[ks]; });
    relu_pack_store<NB>(acc, p, nullptr, hf);
    // f = last layer of semantic_features (no activation) -> wave-private tile, row-major
    chain_layer<NB, KS>(acc, [&](int m, int ks) { return ff[(fl + (size_t)m * KS + ks) * 64 + lane]; }, [&](int ks) { return p[ks]; });
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(LDS_VEC(u32x2)*)(ftile + c * PT + 32 * m + 8 * q + 4 * hf) =
            (u32x2){cvt_pk(acc[m][4 * q], acc[m][4 * q + 1]), cvt_pk(acc[m][4 * q + 2], acc[m][4 * q + 3])};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (SUMS) {   // (a') sum_s w_s f_s of the tile: lanes 0..31 hold rows 0 (head) and 1 (remainder) of the product
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x16 sa = zero16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) sa = mfma16(aw[ks], tr_frag(ftv, 32 * nb, ks, lane), sa);
        if (hf == 0) tile_sums[(size_t)tile * 96 + 32 * nb + c] = sa[0] + sa[1];
      }
    } else {
    // (a) the 32 rows of f leave as contiguous 16-byte pieces
    const int rows_here = min(32, rows - row0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = lane + 64 * i, r = pc >> 3, ch = pc & 7;
      if (r < rows_here) *(u32x4*)(feat + (size_t)(row0 + r) * D + 8 * ch) = *(const LDS_VEC(u32x4)*)(ftile + r * PT + 8 * ch);
    }
    }
    // (b) semantic_out layer 0: relu(f) in natural k-order from the tile, then the geo_feat chunk
    chain_layer<NB, KS0O>(acc, [&](int m, int ks) { return fo_[((size_t)m * KS0O + ks) * 64 + lane]; },
                          [&](int ks) {
                            if (ks >= D / 16) return geo[ks - D / 16];
                            union { u32x4 u; s16x2 i[4]; h16x8 v; } b;
                            b.u = *(const LDS_VEC(u32x4)*)(ftile + c * PT + 16 * ks + 8 * hf);
#pragma unroll
                            for (int j = 0; j < 4; ++j) b.i[j] = __builtin_elementwise_max(b.i[j], (s16x2){0, 0});
                            return b.v;
                          });
    relu_pack_store<NB>(acc, p, nullptr, hf);
    for (int ob = 0; ob < OBO; ++ob) {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) o = mfma16(fo_[(flo + (size_t)ob * KS + ks) * 64 + lane], p[ks], o);
      if constexpr (SUMS) {   // (OBO == 1) sum_s w_s logits_s: the rows are the lanes here -- a butterfly over each half's 32 lanes, per class
        const int nreg = out_pad_o / 2;    // classes (r & 3) + 8 (r >> 2) + 4 half, r < out_pad / 2
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q < nreg) {
            float4 t;
            t.x = half_sum32(wl * (float)(h16)o[4 * q]); t.y = half_sum32(wl * (float)(h16)o[4 * q + 1]);
            t.z = half_sum32(wl * (float)(h16)o[4 * q + 2]); t.w = half_sum32(wl * (float)(h16)o[4 * q + 3]);
            if (c == 0) *(float4*)(tile_sums + (size_t)tile * 96 + 64 + 8 * q + 4 * hf) = t;
          }
        }
      } else if (valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int f = 32 * ob + 8 * q + 4 * hf;
          if (f < out_pad_o) {
            h16x4 v; v[0] = (h16)o[4 * q]; v[1] = (h16)o[4 * q + 1]; v[2] = (h16)o[4 * q + 2]; v[3] = (h16)o[4 * q + 3];
            *(h16x4*)(logits + (size_t)row * out_pad_o + f) = v;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the tile is rewritten by the next iteration's f
  }
}

// ---------------------------------------------------------------- backward (data path)
template <int NB>
__device__ inline bool mask_pack_store(f32x16 (&acc)[NB], h16x8 (&p)[2 * NB], const h16* act_row, h16* dst_row, int hf,
                                       bool valid) {
  bool bad = false;
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      h16x4 a = {0, 0, 0, 0};
      if (valid) a = *(const h16x4*)(act_row + 32 * m + 8 * q + 4 * hf);
      h16x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float g = ((float)a[r] > 0.f) ? acc[m][4 * q + r] : 0.f;
        h16 gh = (h16)g;
        bad |= !(fabsf((float)gh) <= 65504.f);
        v[r] = gh;
        p[2 * m + (q >> 1)][4 * (q & 1) + r] = gh;
      }
      if (valid && dst_row) *(h16x4*)(dst_row + 32 * m + 8 * q + 4 * hf) = v;
    }
  }
  return bad;
}

template <int HID, int NHID>
__global__ __launch_bounds__(256) void k_mlp_bwd(const h16* __restrict__ wb_g, size_t wb_halves, int in_pad, int out_pad,
                                                const h16* __restrict__ h1, const h16* __restrict__ h2,
                                                const h16* __restrict__ d_out, int rows, const int* __restrict__ rows_dev,
                                                h16* __restrict__ dA1, h16* __restrict__ dA2, h16* __restrict__ d_in,
                                                int* __restrict__ found_inf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* wl = (h16*)smem;
  copy_to_lds(wl, wb_g, wb_halves);
  __syncthreads();
  constexpr int NB = HID / 32, KS = HID / 16;
  const h16x8* frag = (const h16x8*)wl;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, c = lane & 31;
  if (rows_dev) rows = min(rows, *rows_dev);
  const int KSO = out_pad / 16, IB = ceil32(in_pad);
  const size_t f1 = (size_t)NB * KSO;
  const size_t fl = f1 + (NHID == 2 ? (size_t)NB * KS : 0);
  const int ntiles = (rows + 31) / 32;
  bool bad = false;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row = tile * 32 + c;
    const bool valid = row < rows;
    f32x16 acc[NB];
    zero_acc(acc);
    for (int ks = 0; ks < KSO; ++ks) {
      h16x8 b = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid) b = *(const h16x8*)(d_out + (size_t)row * out_pad + 16 * ks + 8 * hf);
#pragma unroll
      for (int m = 0; m < NB; ++m) acc[m] = mfma16(frag[((size_t)m * KSO + ks) * 64 + lane], b, acc[m]);
    }
    h16x8 p[KS];
    const h16* hl = (NHID == 2) ? h2 : h1;
    h16* dAl = (NHID == 2) ? dA2 : dA1;
    bad |= mask_pack_store<NB>(acc, p, hl + (size_t)row * HID, dAl ? dAl + (size_t)row * HID : nullptr, hf, valid);
    if (NHID == 2) {
      zero_acc(acc);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < NB; ++m) acc[m] = mfma16(frag[(f1 + (size_t)m * KS + ks) * 64 + lane], p[ks], acc[m]);
      bad |= mask_pack_store<NB>(acc, p, h1 + (size_t)row * HID, dA1 ? dA1 + (size_t)row * HID : nullptr, hf, valid);
    }
    if (d_in) {
      for (int ib = 0; ib < IB; ++ib) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) o = mfma16(frag[(fl + (size_t)ib * KS + ks) * 64 + lane], p[ks], o);
        if (valid) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int f = 32 * ib + 8 * q + 4 * hf;
            if (f < in_pad) {
              h16x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] = (h16)o[4 * q + r]; bad |= !(fabsf((float)v[r]) <= 65504.f); }
              *(h16x4*)(d_in + (size_t)row * in_pad + f) = v;
            }
          }
        }
      }
    }
  }
  if (found_inf && __any(bad) && lane == 0) atomicOr(found_inf, 1);
}

// ---------------------------------------------------------------- weight gradients
// dW[o][i] += sum_r dA[r][o] * X[r][i]   (contraction over rows).  A/B operands need 8 consecutive ROWS per
// lane, i.e. a transposed read of the row-major tiles: staged in LDS, gathered with 16-bit reads.
// HBM-bound (streams dA and X once); superseded by the fused backward once that lands.
#define DW_ROWS 64
__global__ __launch_bounds__(256) void k_dw_gemm(const h16* __restrict__ dA, int OW, const h16* __restrict__ X, int IW,
                                                int rows, const int* __restrict__ rows_dev, float* __restrict__ dW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int OWp = ceil32(OW) * 32, IWp = ceil32(IW) * 32;
  h16* tA = (h16*)smem;               // [DW_ROWS][OWp]
  h16* tX = tA + DW_ROWS * OWp;       // [DW_ROWS][IWp]
  if (rows_dev) rows = min(rows, *rows_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, c = lane & 31;
  const int nob = OWp / 32, nib = IWp / 32, nblk = nob * nib;
  f32x16 acc[4];
  zero_acc(acc);
  for (int i = threadIdx.x; i < DW_ROWS * (OWp + IWp); i += 256) tA[i] = (h16)0.f;
  const int nchunks = (rows + DW_ROWS - 1) / DW_ROWS;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    __syncthreads();
    const int r0 = ch * DW_ROWS;
    for (int i = threadIdx.x; i < DW_ROWS * (OW / 8); i += 256) {
      int r = i / (OW / 8), k = i % (OW / 8);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r0 + r < rows) v = *(const uint4*)(dA + (size_t)(r0 + r) * OW + 8 * k);
      *(uint4*)(tA + r * OWp + 8 * k) = v;
    }
    for (int i = threadIdx.x; i < DW_ROWS * (IW / 8); i += 256) {
      int r = i / (IW / 8), k = i % (IW / 8);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r0 + r < rows) v = *(const uint4*)(X + (size_t)(r0 + r) * IW + 8 * k);
      *(uint4*)(tX + r * IWp + 8 * k) = v;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int blk = wave + 4 * b;
      if (blk < nblk) {
        int ob = blk / nib, ib = blk % nib;
#pragma unroll
        for (int ks = 0; ks < DW_ROWS / 16; ++ks) {
          h16x8 a, bb;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a[j] = tA[(16 * ks + 8 * hf + j) * OWp + 32 * ob + c];
            bb[j] = tX[(16 * ks + 8 * hf + j) * IWp + 32 * ib + c];
          }
          acc[b] = mfma16(a, bb, acc[b]);
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int blk = wave + 4 * b;
    if (blk < nblk) {
      int ob = blk / nib, ib = blk % nib;
      int i = 32 * ib + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int o = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (o < OW && i < IW && acc[b][r] != 0.f) unsafeAtomicAdd(dW + (size_t)o * IW + i, acc[b][r]);
      }
    }
  }
}

// ---------------------------------------------------------------- weight-gradient operands of the fused backward
//   dW_l[o][i] = sum_samples dA_l[s][o] * X_l[s][i]
// The contraction index is the SAMPLE, i.e. both MFMA operands need 8 consecutive samples per lane.  The backward kernel parks
// dA_l and X_l of its 128 samples in LDS as plain row-major [sample][feature] tiles and reads operand fragments with the gfx950
// hardware transpose read ds_read_b64_tr_b16 (16 lanes fetch a 4-sample x 16-feature block; lane i receives feature i of 4
// consecutive samples -- semantics verified by scripts/dev/probe_trread.hip).  dW accumulators live in registers for the whole
// kernel (each dW wave owns a fixed subset of 32x32 C-blocks) and leave once at the end.
// dW accumulate over the block tile: wave owns C-blocks blk = wave + 4b (blk -> (ob, ib) = (blk / NIB, blk % NIB)).
// When 4 % NIB == 0 the input block ib is the same for all owned blocks, so its fragment is fetched once per k-step.
template <int NBLK, int NOB, int NIB, int TROWS = 128, class TVA, class TVB>
__device__ inline void dw_accumulate(f32x16 (&dw)[NBLK], TVA tA, TVB tB, int wave, int lane) {
  constexpr bool IB_CONST = (4 % NIB) == 0;
  constexpr int NBF = IB_CONST ? 1 : NBLK;
  h16x8 a[2][NBLK], bq[2][NBF];
  auto fetch = [&](int ks, int slot) {
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int blk = wave + 4 * b;
      if (blk < NOB * NIB) {
        a[slot][b] = tr_frag(tA, 32 * (blk / NIB), ks, lane);
        if (!IB_CONST) bq[slot][b < NBF ? b : 0] = tr_frag(tB, 32 * (blk % NIB), ks, lane);
      }
    }
    if (IB_CONST) bq[slot][0] = tr_frag(tB, 32 * (wave % NIB), ks, lane);
  };
  fetch(0, 0);
#pragma unroll
  for (int ks = 0; ks < TROWS / 16; ++ks) {
    if (ks + 1 < TROWS / 16) fetch(ks + 1, (ks + 1) & 1);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int blk = wave + 4 * b;
      if (blk < NOB * NIB) dw[b] = mfma16(a[ks & 1][b], bq[ks & 1][IB_CONST ? 0 : (b < NBF ? b : 0)], dw[b]);
    }
  }
}

// relu'(h) mask read back from the LDS tile holding h (C-layout 8-byte chunks)
// fp16 overflow detection without converting every value back: track max |g| (overflow <=> > 65504) and fold the values
// into a NaN catcher (g * 0 is NaN for NaN / inf); two VALU ops per element instead of four.
struct InfTrack {
  float mx = 0.f, nanz = 0.f;
  __device__ inline void see(float g) { mx = fmaxf(mx, fabsf(g)); nanz = fmaf(g, 0.f, nanz); }
  __device__ inline bool bad() const { return !(mx <= 65504.f) || !(nanz == 0.f); }
};
// The recompute kernels do not watch the intermediate fp16 gradients for overflow: an inf in dA either dies under the
// ReLU mask (no effect, as in torch) or reaches dW / d_in as inf / NaN, and those endpoints are checked (GradScaler only
// ever inspects parameter gradients: torch/amp/grad_scaler.py _unscale_grads_).
template <int NB, class TV>
__device__ inline void mask_pack_lds(f32x16 (&acc)[NB], h16x8 (&p)[2 * NB], TV t, int srow, int hf) {
  u32x2 a[NB][4];   // all activation reads go out first (one LDS latency for the layer, not one per 8 bytes)
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) a[m][q] = *(const LDS_VEC(u32x2)*)t.at(srow, 32 * m + 8 * q + 4 * hf);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t* w = (uint32_t*)&p[2 * m + (q >> 1)];
      w[2 * (q & 1)] = mask2(acc[m][4 * q], acc[m][4 * q + 1], a[m][q].x);
      w[2 * (q & 1) + 1] = mask2(acc[m][4 * q + 2], acc[m][4 * q + 3], a[m][q].y);
    }
}

// ---------------------------------------------------------------- fused backward WITH forward recompute
// Nothing but the layer input x and dL/dout is read from HBM: the hidden activations are recomputed in registers
// (forward chain) and parked in LDS only as operands of the weight-gradient MFMAs.  HBM traffic per sample drops from
// 2*HID*NHID + IN + OUT halves read to IN + OUT (sigma head: 770 -> 224 B), which turns the kernel from latency/HBM bound
// into MFMA bound.  Weights live in LDS ONCE, row-major: forward fragments are plain reads, transposed (backward)
// fragments come from ds_read_b64_tr_b16.
// forward A fragment (lane: output feature n = row, 8 input features) from the row-major weights
__device__ inline h16x8 fwd_frag_natural(const lds_h16* W, int pitch, int mb, int ks, int lane) {
  return *(const LDS_VEC(h16x8)*)(W + (32 * mb + (lane & 31)) * pitch + 16 * ks + 8 * (lane >> 5));
}
// chained k-order: two 8-byte pieces 8 halves apart, read as two ds_read_b64 (never one ds_read2_b64: 8+ LDS cycles and
// 32-bank conflicts against 2 x 2 conflict-free cycles)
template <class TV>
__device__ inline h16x8 fwd_frag_chained(TV W, int mb, int ks, int lane) {
  const int row = 32 * mb + (lane & 31), col = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * (lane >> 5);
  union { struct { u32x2 a, b; } s; h16x8 v; } u;
  auto p0 = W.at(row, col);
  auto p1 = W.at(row, col + 8);
  asm volatile("" : "+v"(p1));   // opaque: keeps the two reads from being fused into one ds_read2_b64
  u.s.a = *(const LDS_VEC(u32x2)*)p0;
  u.s.b = *(const LDS_VEC(u32x2)*)p1;
  return u.v;
}
template <int NB, class TV>
__device__ inline void write_packed_tile(TV t, int srow, const h16x8 (&p)[2 * NB], int hf) {
#pragma unroll
  for (int m = 0; m < NB; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t* w = (const uint32_t*)&p[2 * m + (q >> 1)];
      *(LDS_VEC(u32x2)*)t.at(srow, 32 * m + 8 * q + 4 * hf) = (u32x2){w[2 * (q & 1)], w[2 * (q & 1) + 1]};
    }
}
// ---------------------------------------------------------------- recompute backward, wave-specialised (8 waves)
// Profiling the first version (4 waves, every wave chain + dW; rocprofv3 + ISA): ~6500 issued instructions per 128-row tile and wave, of which 152 are
// MFMAs -- with 428 registers per wave the accumulators spill into the AGPR half (1900 v_accvgpr moves) and only ONE wave
// per SIMD is resident, so nothing hides the dependent-issue latency.  Here the two jobs get their own waves: waves 0-3 run
// the register chain (forward recompute + backward data path, ~170 VGPRs), waves 4-7 only accumulate the weight gradients
// (112 accumulator registers + fragments).  Both fit 256 registers, so the block runs 2 waves per SIMD and the chain wave
// and the dW wave of a SIMD overlap their MFMA / LDS latencies.
// Dev-only phase timing (scripts/dev/probe_bwd_phases.py builds with -DALN_PHASE_TIMING): block 0 accumulates the shader
// clock spent between consecutive stamps, per role (0 = chain wave 0, 1 = dW wave 4).
#ifdef ALN_PHASE_TIMING
__device__ long long g_phase_cycles[2][32];
__device__ int g_phase_in = 0;    // 0: every instantiation records; else only the one with this input width
extern "C" int aln_debug_read_phases(long long* host_out, int reset) {
  if (reset) {
    long long z[64] = {0};
    const int sel = reset > 1 ? reset : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_in), &sel, sizeof(int));
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles), sizeof(long long) * 64);
}
#define PT_DECL long long pt_acc[32] = {0}; long long pt_last = clock64();
#define PT_STAMP(i) { long long pt_now = clock64(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; }
#define PT_FLUSH(role) if (blockIdx.x == 0 && cw == 0 && lane == 0 && (g_phase_in == 0 || g_phase_in == IN)) { for (int i = 0; i < 32; ++i) g_phase_cycles[role][i] += pt_acc[i]; }
#else
#define PT_DECL
#define PT_STAMP(i)
#define PT_FLUSH(role)
#endif

// The x / dOut chunks of the NEXT tile are requested right after B1 (raw_load: XM / DM are the row-source modes) and sit in
// registers through the whole tile, so the global-load latency (15 % of a tile when exposed; more for the on-the-fly
// semantic-head sources) hides behind the MFMA phases; raw_finish + the LDS stores run at the top of the next tile.
// OCC2 (64-wide heads whose tiles fit 80 KB): two blocks per CU at <= 128 VGPRs.  These heads carry ~1000 MFMA cycles per tile
// against ~9000 cycles of barrier / LDS / load latency, so a second resident block fills the stalls of the first; the cross-tile
// prefetch registers are dropped (the other block hides the load latency) and the x tile is stored at pitch IN + 8 (the
// weight-gradient blocks read up to 32 * IB columns: what they pick up past IN belongs to columns that are never flushed).
template <int IN, int HID, int OUT, int NHID, int XM, int DM, bool OCC2, int RPW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC2 ? 4 : 2, OCC2 ? 4 : 2))) void k_mlp_bwd_recomp8(const h16* __restrict__ wr_g, size_t wr_halves, RowSrc xs, RowSrc ds,
                                                        int rows, const int* __restrict__ rows_dev, h16* __restrict__ d_in,
                                                        float* __restrict__ dW, float* __restrict__ dw_ws,
                                                        int* __restrict__ found_inf) {
  constexpr int NB = HID / 32, KS = HID / 16, KS0 = IN / 16, KSO = OUT / 16, IB = (IN + 31) / 32, OB = (OUT + 31) / 32;
  constexpr int PW0 = IN + 8, PW1 = hid_pitch(HID);
  constexpr int PH = hid_pitch(HID), PX0 = OCC2 ? IN + 8 : IB * 32 + 8, PO = OB * 32 + (OB == 1 ? 0 : 8);   // 32-wide dOut tile: transpose reads conflict-free unpadded
  constexpr int TR = 128 * RPW;     // RPW 32-row sub-tiles per chain wave (64-wide heads: 2, the phases are latency-bound)
  constexpr int NBLK_LAST = (OB * NB + 3) / 4, NBLK_MID = (NB * NB + 3) / 4, NBLK_FIRST = (NB * IB + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_h16* wl = (lds_h16*)smem;
  lds_h16* W0 = wl;
  lds_h16* W1 = W0 + HID * PW0;
  lds_h16* WL = (NHID == 2) ? W1 + HID * PW1 : W1;
  lds_h16* tX0 = wl + (int)((wr_halves + 7) & ~(size_t)7);
  lds_h16* b1 = tX0 + TR * PX0;
  lds_h16* b2 = b1 + TR * PH;
  lds_h16* tO = b2 + TR * PH;
  const PlainV<lds_h16*> vW0{W0, PW0}, vW1{W1, PW1}, vWL{WL, PW1}, vX0{tX0, PX0}, vb1{b1, PH}, vb2{b2, PH}, vO{tO, PO};
  copy_to_lds((h16*)smem, wr_g, wr_halves);
  for (int i = threadIdx.x; i < TR * (PX0 + PO); i += 512) { if (i < TR * PX0) tX0[i] = (h16)0.f; else tO[i - TR * PX0] = (h16)0.f; }
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool chain = wave < 4;
  const int cw = wave & 3;                     // chain waves: 32-row slice ; dW waves: C-block owner id
  const int srow0 = cw * 32 * RPW + c;   // sub-tile u of this chain wave: rows srow0 + 32 u
  if (rows_dev) rows = min(rows, *rows_dev);
  const int ntiles = (rows + TR - 1) / TR;
  constexpr int XCH = IN / 8, OCH = OUT / 8;                       // 8-half chunks per row
  constexpr int NXS = (TR * XCH + 511) / 512, NOS = (TR * OCH + 511) / 512;
  constexpr bool LG = RPW > 1;   // 256-row tiles: twice the prefetch registers, so the cache-resident per-ray rows are read late
  // item i of the x tile -> (row, 8-half chunk).  semantic_out's input is [relu(f) (IN - 16) | geo_feat, 1 (16)]: its two kinds of
  // chunks come from different tensors through different code, so the f chunks of the whole tile are numbered first and the geo
  // chunks after them -- every wave then builds one kind only (mixed, both branches ran in every wave: 22 us of a 165 us launch)
  auto xmap = [](int i, int& r, int& k) {
    if constexpr (XM == SRC_SEMO_IN) {
      constexpr int FCH = XCH - 2;
      if (i < TR * FCH) { r = i / FCH; k = i % FCH; }
      else { const int j = i - TR * FCH; r = j >> 1; k = FCH + (j & 1); }
    } else { r = i / XCH; k = i % XCH; }
  };
  // Per-ray output gradients (DLOGITS / DSEMF_OUT sources): the rows of the <= GT_RAYS rays a tile spans are staged in LDS one
  // tile ahead by an otherwise idle weight-gradient wave, so building a 16-byte chunk costs two LDS reads instead of a trip to
  // L2 per chunk (eight chunks per sample row for the 64-wide feature gradient).  Tiles that span more rays fall back to global loads.
  constexpr bool GT = DM == SRC_DLOGITS || DM == SRC_DSEMF_OUT;
  constexpr int GTW = OB * 32;                 // floats per staged ray row (>= OUT, zero beyond gw)
  float* const gt = (float*)(tO + TR * PO);    // [2 buffers][GT_RAYS][GTW]
  auto gt_fill = [&](int r0n, int buf) {       // called by ONE wave
    if constexpr (GT) {
      const int first = row_ray(ds, r0n);
      for (int e = 4 * lane; e < GT_RAYS * GTW; e += 256) {
        const int ray = first + e / GTW, c = e % GTW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ray < ds.N) {
          const float* g = ds.g + (size_t)ray * ds.gw + c;
          if ((ds.gw & 3) == 0 && c + 3 < ds.gw) v = *(const float4*)g;
          else { if (c < ds.gw) v.x = g[0]; if (c + 1 < ds.gw) v.y = g[1]; if (c + 2 < ds.gw) v.z = g[2]; if (c + 3 < ds.gw) v.w = g[3]; }
        }
        *(float4*)(gt + buf * (GT_RAYS * GTW) + e) = v;
      }
    }
  };
  RawChunk px[OCC2 ? 1 : NXS], po[OCC2 ? 1 : NOS];
  auto prefetch = [&](int r0) {
    if constexpr (OCC2) return;
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      const int i = threadIdx.x + 512 * q; int r, k; xmap(i, r, k);
      if (i < TR * XCH && r0 + r < rows) raw_load<XM, LG>(px[q], xs, r0 + r, 8 * k);
    }
#pragma unroll
    for (int q = 0; q < NOS; ++q) {
      const int i = threadIdx.x + 512 * q, r = i / OCH, k = i % OCH;
      if (i < TR * OCH && r0 + r < rows) raw_load<DM, LG || GT>(po[q], ds, r0 + r, 8 * k);
    }
  };
  auto stash = [&](int r0, int gbuf) {
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* const gtb = GT ? gt + gbuf * (GT_RAYS * GTW) : nullptr;
    const int gfirst = GT ? row_ray(ds, r0) : 0;
    if constexpr (OCC2) {   // load and store in one go: nothing of the tile stays in registers
#pragma unroll
      for (int q = 0; q < NXS; ++q) {
        const int i = threadIdx.x + 512 * q; int r, k; xmap(i, r, k);
        if (i < TR * XCH) {
          h16x8 v = z;
          if (r0 + r < rows) { RawChunk t; raw_load<XM>(t, xs, r0 + r, 8 * k); v = raw_finish<XM>(t, xs, 8 * k); }
          *(LDS_VEC(h16x8)*)(tX0 + r * PX0 + 8 * k) = v;
        }
      }
#pragma unroll
      for (int q = 0; q < NOS; ++q) {
        const int i = threadIdx.x + 512 * q, r = i / OCH, k = i % OCH;
        if (i < TR * OCH) {
          h16x8 v = z;
          if (r0 + r < rows) { RawChunk t; raw_load<DM, GT>(t, ds, r0 + r, 8 * k); v = raw_finish<DM, GT>(t, ds, 8 * k, gtb, gfirst, GTW); }
          *(LDS_VEC(h16x8)*)(tO + r * PO + 8 * k) = v;
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      const int i = threadIdx.x + 512 * q; int r, k; xmap(i, r, k);
      if (i < TR * XCH) *(LDS_VEC(h16x8)*)(tX0 + r * PX0 + 8 * k) = (r0 + r < rows) ? raw_finish<XM, LG>(px[q], xs, 8 * k) : z;
    }
#pragma unroll
    for (int q = 0; q < NOS; ++q) {
      const int i = threadIdx.x + 512 * q, r = i / OCH, k = i % OCH;
      if (i < TR * OCH) *(LDS_VEC(h16x8)*)(tO + r * PO + 8 * k) = (r0 + r < rows) ? raw_finish<DM, LG || GT>(po[q], ds, 8 * k, gtb, gfirst, GTW) : z;
    }
  };
  if ((int)blockIdx.x < ntiles) { prefetch(blockIdx.x * TR); if (wave == 4) gt_fill(blockIdx.x * TR, 0); }
  // The two roles run SEPARATE tile loops with the same barrier sequence (s_barrier only counts arrivals, and the role is
  // wave-uniform), so the register allocator never sees the chain state and the dW accumulators live at the same time.
  if (chain) {
    h16x2 nanz = {0, 0};
    PT_DECL
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
      const int r0 = tile * TR;
      PT_STAMP(0) __syncthreads(); PT_STAMP(1)   // B0
      stash(r0, it & 1);
      PT_STAMP(2) __syncthreads(); PT_STAMP(3)   // B1
      { const int nt = tile + gridDim.x; if (nt < ntiles) prefetch(nt * TR); }
      f32x16 acc[RPW][NB];
      h16x8 p[RPW][KS];
      chain_layer_m<RPW, NB, KS0>(acc, [&](int m, int ks) { return fwd_frag_natural(W0, PW0, m, ks, lane); },
                                  [&](int u, int ks) { return *(const LDS_VEC(h16x8)*)(tX0 + (srow0 + 32 * u) * PX0 + 16 * ks + 8 * hf); });
      PT_STAMP(16)
#pragma unroll
      for (int u = 0; u < RPW; ++u) relu_pack_store<NB>(acc[u], p[u], nullptr, hf);
      PT_STAMP(17)
#pragma unroll
      for (int u = 0; u < RPW; ++u) write_packed_tile<NB>(vb1, srow0 + 32 * u, p[u], hf);
      PT_STAMP(18)
      if constexpr (NHID == 2) {
        chain_layer_m<RPW, NB, KS>(acc, [&](int m, int ks) { return fwd_frag_chained(vW1, m, ks, lane); }, [&](int u, int ks) { return p[u][ks]; });
        PT_STAMP(19)
#pragma unroll
        for (int u = 0; u < RPW; ++u) relu_pack_store<NB>(acc[u], p[u], nullptr, hf);
        PT_STAMP(20)
#pragma unroll
        for (int u = 0; u < RPW; ++u) write_packed_tile<NB>(vb2, srow0 + 32 * u, p[u], hf);
      }
      PT_STAMP(4) __syncthreads(); PT_STAMP(5)   // B2
      chain_layer_m<RPW, NB, KSO>(acc, [&](int m, int ks) { return tr_frag(vWL, 32 * m, ks, lane); },
                                  [&](int u, int ks) { return *(const LDS_VEC(h16x8)*)(tO + (srow0 + 32 * u) * PO + 16 * ks + 8 * hf); });
#pragma unroll
      for (int u = 0; u < RPW; ++u) { if constexpr (NHID == 2) mask_pack_lds<NB>(acc[u], p[u], vb2, srow0 + 32 * u, hf); else mask_pack_lds<NB>(acc[u], p[u], vb1, srow0 + 32 * u, hf); }
      PT_STAMP(6) __syncthreads(); PT_STAMP(7)   // B3
      if constexpr (NHID == 2) {
#pragma unroll
        for (int u = 0; u < RPW; ++u) write_packed_tile<NB>(vb2, srow0 + 32 * u, p[u], hf);                      // dA2 over h2
        PT_STAMP(8) __syncthreads(); PT_STAMP(9)   // B4
        chain_layer_m<RPW, NB, KS>(acc, [&](int m, int ks) { return tr_frag_chained(vW1, 32 * m, ks, lane); }, [&](int u, int ks) { return p[u][ks]; });
        PT_STAMP(21)
#pragma unroll
        for (int u = 0; u < RPW; ++u) mask_pack_lds<NB>(acc[u], p[u], vb1, srow0 + 32 * u, hf);
        PT_STAMP(10) __syncthreads(); PT_STAMP(11)   // B5
      }
#pragma unroll
      for (int u = 0; u < RPW; ++u) write_packed_tile<NB>(vb2, srow0 + 32 * u, p[u], hf);                        // dA1
      PT_STAMP(12) __syncthreads(); PT_STAMP(13)   // B6
      if (d_in) {
        constexpr int UJ = (IB > 1) ? 1 : RPW;     // wide inputs: one sub-tile at a time (RPW x IB x 16 accumulators would not fit)
        constexpr bool FOLD = DM == SRC_DSEMF_OUT && IN == 16;
        h16x4 skip[RPW][2] = {};
        if constexpr (FOLD) {   // requested ahead of the MFMAs below: the lines were just read by the tile load, L2 hits
          if (ds.fold_geo) {
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
              const int row = r0 + srow0 + 32 * u;
              if (row < rows) {
#pragma unroll
                for (int q = 0; q < 2; ++q) skip[u][q] = *(const h16x4*)(ds.b + (size_t)row * ds.ldb + ds.D + 8 * q + 4 * hf);
              }
            }
          }
        }
        f32x16 o[UJ][IB];
        if constexpr (UJ == RPW) chain_layer_m<RPW, IB, KS>(o, [&](int ib, int ks) { return tr_frag_chained(vW0, 32 * ib, ks, lane); }, [&](int u, int ks) { return p[u][ks]; });
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
          if constexpr (UJ != RPW) chain_layer_m<1, IB, KS>(o, [&](int ib, int ks) { return tr_frag_chained(vW0, 32 * ib, ks, lane); }, [&](int, int ks) { return p[u][ks]; });
          const int row = r0 + srow0 + 32 * u;
          if (row < rows) {
#pragma unroll
            for (int ib = 0; ib < IB; ++ib)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int f = 32 * ib + 8 * q + 4 * hf;
                if (f < IN) {
                  h16x4 v;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    float ov = o[UJ == RPW ? u : 0][ib][4 * q + r];
                    if constexpr (FOLD) ov += (float)skip[u][q < 2 ? q : 0][r];
                    v[r] = (h16)ov;
                  }
                  if constexpr (XM == SRC_SEMO_IN) {   // columns [0, D) are relu(f): hand dL/df on (the x tile holds relu(f))
                    if (f < xs.D) {
                      const h16x4 xm = *(const LDS_VEC(h16x4)*)(tX0 + (srow0 + 32 * u) * PX0 + f);
#pragma unroll
                      for (int r = 0; r < 4; ++r) v[r] = (float)xm[r] > 0.f ? v[r] : (h16)0.f;
                    }
                  }
                  nanz = nan_fold((h16x2){v[0], v[1]}, nan_fold((h16x2){v[2], v[3]}, nanz));
                  *(h16x4*)(d_in + (size_t)row * IN + f) = v;
                }
              }
          }
        }
      }
    }
    PT_FLUSH(0)
    if (found_inf && __any(nan_bad(nanz)) && lane == 0) atomicOr(found_inf, 1);
  } else {
    MlpLayers L = mlp_layers(IN, HID, OUT, NHID);
    f32x16 dw_last[NBLK_LAST], dw_mid[NHID == 2 ? NBLK_MID : 1], dw_first[NBLK_FIRST];
    zero_acc(dw_last); zero_acc(dw_mid); zero_acc(dw_first);
    PT_DECL
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
      const int r0 = tile * TR;
      PT_STAMP(0) __syncthreads(); PT_STAMP(1)   // B0
      stash(r0, it & 1);
      PT_STAMP(2) __syncthreads(); PT_STAMP(3)   // B1
      { const int nt = tile + gridDim.x; if (nt < ntiles) { prefetch(nt * TR); if (wave == 4) gt_fill(nt * TR, (it + 1) & 1); } }
      PT_STAMP(4) __syncthreads(); PT_STAMP(5)   // B2
      if constexpr (NHID == 2) dw_accumulate<NBLK_LAST, OB, NB, TR>(dw_last, vO, vb2, cw, lane); else dw_accumulate<NBLK_LAST, OB, NB, TR>(dw_last, vO, vb1, cw, lane);
      PT_STAMP(6) __syncthreads(); PT_STAMP(7)   // B3
      if constexpr (NHID == 2) {
        PT_STAMP(8) __syncthreads(); PT_STAMP(9)   // B4
        dw_accumulate<NBLK_MID, NB, NB, TR>(dw_mid, vb2, vb1, cw, lane);
        PT_STAMP(10) __syncthreads(); PT_STAMP(11)   // B5
      }
      PT_STAMP(12) __syncthreads(); PT_STAMP(13)   // B6
      dw_accumulate<NBLK_FIRST, NB, IB, TR>(dw_first, vb2, vX0, cw, lane);
    }
    PT_STAMP(0)
    PT_FLUSH(1)
    if (dW) {
      bool bad = false;
      // The block's sums leave as plain coalesced stores into its slab of the partial-sum workspace (dw_ws: one [n_weights] slab
      // per block) and k_dw_reduce adds the slabs up in a fixed order: no atomics, bit-reproducible weight gradients.  Measured before:
      // 256-512 blocks finishing together and adding into the same 9-25 K floats with fp32 atomics cost ~0.1 us per block
      // (27 us of a 115 us launch for the 64-wide heads) -- every line takes one add per block, serialised at the memory side.
      float* const slab = dw_ws + (size_t)blockIdx.x * (L.w_off[L.n - 1] + (size_t)L.in_[L.n - 1] * L.out_[L.n - 1]);
      auto flush = [&](f32x16& a, int ob, int ib, int OUTL, int INL, size_t off) {
        const int i = 32 * ib + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * hf;
          const bool in_w = o < OUTL && i < INL;
          bad |= in_w && !(fabsf(a[r]) <= 3.0e38f);
          if (in_w) slab[off + (size_t)o * INL + i] = a[r];
        }
      };
#pragma unroll
      for (int b = 0; b < NBLK_LAST; ++b) { int blk = cw + 4 * b; if (blk < OB * NB) flush(dw_last[b], blk / NB, blk % NB, OUT, HID, L.w_off[L.n - 1]); }
      if constexpr (NHID == 2) {
#pragma unroll
        for (int b = 0; b < NBLK_MID; ++b) { int blk = cw + 4 * b; if (blk < NB * NB) flush(dw_mid[b], blk / NB, blk % NB, HID, HID, L.w_off[1]); }
      }
#pragma unroll
      for (int b = 0; b < NBLK_FIRST; ++b) { int blk = cw + 4 * b; if (blk < NB * IB) flush(dw_first[b], blk / IB, blk % IB, HID, IN, L.w_off[0]); }
      if (found_inf && __any(bad) && lane == 0) atomicOr(found_inf, 1);
    }
  }
}

// dW[e] += sum over the blocks' slabs, fixed order: 16 interleaved partial sums per element (thread (e, g) adds slabs g, g + 16, ...
// with all its loads in flight at once), folded pairwise in LDS.  (4 partial sums and 256 elements per block took 6.5 us per
// head, all of it load latency: 64 dependent-in-batches loads per thread.)
#define DWR_G 16
#define DWR_E 64
__global__ __launch_bounds__(DWR_G * DWR_E) void k_dw_reduce(const float* __restrict__ ws, int nparts, int n, float* __restrict__ dW) {
  __shared__ float part[DWR_G][DWR_E];
  const int t = threadIdx.x % DWR_E, g = threadIdx.x / DWR_E, e = blockIdx.x * DWR_E + t;
  float acc = 0.f;
  if (e < n) {
    int p = g;
#pragma unroll 1
    for (; p + 15 * DWR_G < nparts; p += 16 * DWR_G) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ws[(size_t)(p + DWR_G * u) * n + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; p < nparts; p += DWR_G) acc += ws[(size_t)p * n + e];
  }
  part[g][t] = acc;
  __syncthreads();
#pragma unroll
  for (int h = DWR_G / 2; h > 0; h >>= 1) {
    if (g < h) part[g][t] += part[g + h][t];
    __syncthreads();
  }
  if (g == 0 && e < n) dW[e] += part[0][t];
}

// The same for several heads in ONE launch (AlnMlpDesc.defer_dw_reduce: the backward kernels of a training step leave their slabs
// in place and the step reduces all of them at once: one launch instead of one per head).
struct DwReduceAll { int n; const float* ws[ALN_MAX_HEADS]; float* dW[ALN_MAX_HEADS]; int nparts[ALN_MAX_HEADS], nw[ALN_MAX_HEADS], blk0[ALN_MAX_HEADS + 1]; };
__global__ __launch_bounds__(DWR_G * DWR_E) void k_dw_reduce_all(DwReduceAll a) {
  __shared__ float part[DWR_G][DWR_E];
  int h = 0;
  while (h + 1 < a.n && (int)blockIdx.x >= a.blk0[h + 1]) ++h;
  const float* __restrict__ ws = a.ws[h];
  const int n = a.nw[h], nparts = a.nparts[h];
  const int t = threadIdx.x % DWR_E, g = threadIdx.x / DWR_E, e = ((int)blockIdx.x - a.blk0[h]) * DWR_E + t;
  float acc = 0.f;
  if (e < n) {
    int p = g;
#pragma unroll 1
    for (; p + 15 * DWR_G < nparts; p += 16 * DWR_G) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ws[(size_t)(p + DWR_G * u) * n + e];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; p < nparts; p += DWR_G) acc += ws[(size_t)p * n + e];
  }
  part[g][t] = acc;
  __syncthreads();
#pragma unroll
  for (int hh = DWR_G / 2; hh > 0; hh >>= 1) {
    if (g < hh) part[g][t] += part[g + hh][t];
    __syncthreads();
  }
  if (g == 0 && e < n) a.dW[h][e] += part[0][t];
}

// slabs the recompute backward of this head writes for `rows` rows (= its grid size): what aln_mlp_dw_reduce_all adds up
static int bwd_recomp_blocks(const AlnMlpDesc* m, int rows) {
  const int tiles = (rows + 127) / 128;
  bool occ2 = false;
  if (m->hidden == 64 && m->in_pad <= 32) {
    const int OB = (m->out_pad + 31) / 32, PO = OB * 32 + (OB == 1 ? 0 : 8);
    const size_t halves = (size_t)aln_mlp_rowmajor_halves(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
    occ2 = (((halves + 7) & ~(size_t)7) + 128 * (size_t)(m->in_pad + 8 + 2 * hid_pitch(m->hidden) + PO)) * 2 + 2 * GT_RAYS * OB * 32 * sizeof(float) <= 80 * 1024;
  }
  const int gmax = occ2 ? 512 : 256;
  return tiles < gmax ? tiles : gmax;
}
extern "C" int32_t aln_mlp_bwd_blocks(const AlnMlpDesc* m, int32_t rows) { return (m && rows > 0) ? bwd_recomp_blocks(m, rows) : 0; }
static int dw_reduce_all_impl(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, const int32_t* slabs, void* stream);
extern "C" int aln_mlp_dw_reduce_all(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, void* stream) {
  return dw_reduce_all_impl(n_heads, descs, dW, rows, nullptr, stream);
}
// the same with the slab count of each head given (slabs[k] > 0: aln_sem_heads_bwd_slabs; 0: aln_mlp_bwd_blocks(descs[k], rows[k]))
extern "C" int aln_mlp_dw_reduce_slabs(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, const int32_t* slabs,
                                       void* stream) {
  return dw_reduce_all_impl(n_heads, descs, dW, rows, slabs, stream);
}
static int dw_reduce_all_impl(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, const int32_t* slabs, void* stream) {
  ALN_REQUIRE(n_heads >= 0 && n_heads <= ALN_MAX_HEADS && (n_heads == 0 || (descs && dW && rows)), "dw_reduce_all: bad arguments");
  DwReduceAll a; a.n = 0; a.blk0[0] = 0;
  for (int k = 0; k < n_heads; ++k) {
    const AlnMlpDesc* m = descs[k];
    ALN_REQUIRE(m && m->dw_ws && dW[k], "dw_reduce_all: NULL pointer in head %d", k);
    if (rows[k] <= 0) continue;
    const MlpLayers L = mlp_layers(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
    const int n_w = (int)(L.w_off[L.n - 1] + (size_t)L.in_[L.n - 1] * L.out_[L.n - 1]);
    const int parts = (slabs && slabs[k] > 0) ? slabs[k] : bwd_recomp_blocks(m, rows[k]);
    ALN_REQUIRE((size_t)m->dw_ws_bytes >= (size_t)parts * n_w * sizeof(float), "dw_reduce_all: dw_ws of head %d too small", k);
    a.ws[a.n] = (const float*)m->dw_ws; a.dW[a.n] = dW[k]; a.nparts[a.n] = parts; a.nw[a.n] = n_w;
    a.blk0[a.n + 1] = a.blk0[a.n] + (n_w + DWR_E - 1) / DWR_E;
    ++a.n;
  }
  if (a.n == 0) return 0;
  hipLaunchKernelGGL(k_dw_reduce_all, dim3(a.blk0[a.n]), dim3(DWR_G * DWR_E), 0, (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("dw_reduce_all");
  return 0;
}

template <int IN, int HID, int OUT, int NHID, bool OCC2, int RPW>
static int launch_bwd_recomp_occ(const AlnMlpDesc* m, RowSrc xs, RowSrc ds, int rows, const int* rows_dev, void* d_in,
                                 float* dW, int* found_inf, hipStream_t s) {
  constexpr int IB = (IN + 31) / 32, OB = (OUT + 31) / 32;
  constexpr int PH = hid_pitch(HID), PX0 = OCC2 ? IN + 8 : IB * 32 + 8, PO = OB * 32 + (OB == 1 ? 0 : 8);   // as in the kernel
  size_t halves = (size_t)aln_mlp_rowmajor_halves(IN, HID, OUT, NHID);
  constexpr int TR = 128 * RPW;
  const bool gt_src = ds.mode == SRC_DLOGITS || ds.mode == SRC_DSEMF_OUT;
  size_t lds = (((halves + 7) & ~(size_t)7) + TR * (size_t)(PX0 + 2 * PH + PO)) * 2 + (gt_src ? 2 * GT_RAYS * OB * 32 * sizeof(float) : 0);   // + the staged per-ray gradient rows
  ALN_REQUIRE(lds <= 160 * 1024, "mlp_bwd_recomp: LDS %zu B exceeds 160 KiB", lds);
  // the slab count comes from the ONE helper aln_mlp_dw_reduce_all uses as well; it knows 128-row tiles and this file's OCC2 rule
  static_assert(RPW == 1, "bwd_recomp_blocks() assumes 128-row tiles: teach it RPW before instantiating RPW > 1");
  const int g = bwd_recomp_blocks(m, rows);
  ALN_REQUIRE(g == min((rows + TR - 1) / TR, OCC2 ? 512 : 256), "mlp_bwd_recomp: slab count %d disagrees with the launch shape (OCC2 = %d)", g, (int)OCC2);
  const MlpLayers LL = mlp_layers(IN, HID, OUT, NHID);
  const int n_w = (int)(LL.w_off[LL.n - 1] + (size_t)LL.in_[LL.n - 1] * LL.out_[LL.n - 1]);
  ALN_REQUIRE(!dW || (m->dw_ws && (size_t)m->dw_ws_bytes >= (size_t)g * n_w * sizeof(float)),
              "mlp_bwd: AlnMlpDesc.dw_ws must hold %d slabs of %d floats (aln_mlp_dw_ws_bytes)", g, n_w);
  float* ws = dW ? (float*)m->dw_ws : nullptr;
#define LAUNCH_SRC(XM, DM)                                                                                                  \
  do {                                                                                                                     \
    hipFuncSetAttribute((const void*)k_mlp_bwd_recomp8<IN, HID, OUT, NHID, XM, DM, OCC2, RPW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        (int)lds);                                                                                         \
    hipLaunchKernelGGL((k_mlp_bwd_recomp8<IN, HID, OUT, NHID, XM, DM, OCC2, RPW>), dim3(g), dim3(512), lds, s, (const h16*)m->wr, halves, \
                       xs, ds, rows, rows_dev, (h16*)d_in, dW, ws, found_inf);                                             \
  } while (0)
  if (xs.mode == SRC_PLAIN && ds.mode == SRC_PLAIN) LAUNCH_SRC(SRC_PLAIN, SRC_PLAIN);
  else if constexpr (IN == 16 && OUT == 64) {
    ALN_REQUIRE(xs.mode == SRC_SEMF_IN && ds.mode == SRC_DSEMF_OUT, "mlp_bwd_recomp: unsupported row sources %d/%d", xs.mode, ds.mode);
    LAUNCH_SRC(SRC_SEMF_IN, SRC_DSEMF_OUT);
  } else if constexpr (IN == 80) {
    ALN_REQUIRE(xs.mode == SRC_SEMO_IN && ds.mode == SRC_DLOGITS, "mlp_bwd_recomp: unsupported row sources %d/%d", xs.mode, ds.mode);
    LAUNCH_SRC(SRC_SEMO_IN, SRC_DLOGITS);
  } else {
    aln_set_error("mlp_bwd_recomp: row sources %d/%d only exist for the semantic heads", xs.mode, ds.mode);
    return -1;
  }
#undef LAUNCH_SRC
  ALN_CHECK_LAUNCH("mlp_bwd_recomp");
  if (ws && !m->defer_dw_reduce) {
    hipLaunchKernelGGL(k_dw_reduce, dim3((n_w + DWR_E - 1) / DWR_E), dim3(DWR_G * DWR_E), 0, s, ws, g, n_w, dW);
    ALN_CHECK_LAUNCH("dw_reduce");
  }
  return 0;
}
// 64-wide heads with a narrow input (semantic_features): two blocks per CU (OCC2).  Measured in the training step (2^20 rows,
// on-the-fly row sources): one block per CU 172 us, OCC2 150 us.
template <int IN, int HID, int OUT, int NHID>
static int launch_bwd_recomp(const AlnMlpDesc* m, RowSrc xs, RowSrc ds, int rows, const int* rows_dev, void* d_in,
                             float* dW, int* found_inf, hipStream_t s) {
  if constexpr (HID == 64 && IN <= 32) {   // (IN = 80, semantic_out, does not fit the 128-VGPR budget of two blocks per CU: 60-70 registers
                                            //  spill and the kernel goes 165 -> 230 us, measured in round 3)
    constexpr int OB = (OUT + 31) / 32, PO = OB * 32 + (OB == 1 ? 0 : 8);
    const size_t halves = (size_t)aln_mlp_rowmajor_halves(IN, HID, OUT, NHID);
    const size_t lds2 = (((halves + 7) & ~(size_t)7) + 128 * (size_t)(IN + 8 + 2 * hid_pitch(HID) + PO)) * 2 + 2 * GT_RAYS * OB * 32 * sizeof(float);
    if (lds2 <= 80 * 1024) return launch_bwd_recomp_occ<IN, HID, OUT, NHID, true, 1>(m, xs, ds, rows, rows_dev, d_in, dW, found_inf, s);
  }
  return launch_bwd_recomp_occ<IN, HID, OUT, NHID, false, 1>(m, xs, ds, rows, rows_dev, d_in, dW, found_inf, s);
}

// ---------------------------------------------------------------- both semantic heads, backward, ONE kernel
// semantic_features (G -> 64 -> 64 -> 64) and semantic_out (cat[relu(f), geo_feat] -> 64 -> C) of models.py:248-256, forward recompute
// included, from sigma_out, the compositing weights and the per-ray output gradients alone: the two-launch path moves 550 B per
// sample row through HBM (f, d(semantic_out input) written and read back, sigma_out twice) and is half bound by that traffic
// (DESIGN.md 4.6); here 36 B are read and 32 B written per row, and nothing but the matrices' gradients leaves the CU.
// One wave per SIMD, two roles with their own tile loops (the register allocator never sees both states at once):
//   * waves 0-1, the CHAIN: 32 sample rows each through the whole chain of both heads (62 MFMAs) with every weight fragment of both
//     heads resident in registers (62 fragments, 248 registers: what a 64-wide pair allows and the 128-wide heads do not) -- the
//     chain reads nothing from LDS but its own rows (two natural-order read-backs, the ReLU masks);
//   * waves 2-3, the WEIGHT GRADIENTS: whole matrices per wave (W2, W1, W0 | V0, V1), so every transposed fragment read feeds two or
//     three MFMAs; accumulators in the AGPR half for the whole kernel.
// Activations and gradients of a 64-row tile are parked in LDS as row-major [sample][feature] tiles, in two sets: the chain fills
// one while the weight-gradient waves read the other (one barrier per tile).  dL/d(semantic_out input)[:, :64] and dL/df never exist
// outside a CU; d(geo_feat) of both heads leaves as one 16-wide row.
// (First version, measured: four symmetric waves with the backward fragments streamed from L2 every tile -- 11 800 ticks per tile in
//  the chain, all of it exposed load latency: 240 us, no better than the two launches.)
__device__ inline void mfma_acc_a(f32x16& acc, h16x8 a, h16x8 b) {   // accumulator in the AGPR half (see mlp_bwd128.hip: mfma_acc)
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// NOB x NIB weight-gradient blocks of one matrix over the tile's 64 rows: dW[o][i] += sum_s A[s][o] B[s][i]
// (tr_frag_s8: both operands take their samples in the bank-friendly order)
#ifdef ALN_TRS8_OLD   // (dev builds only)
#define tr_frag_s8 tr_frag
#endif
template <int NOB, int NIB, int OFF, int NDW, class TVA, class TVB>
__device__ inline void dw_matrix(f32x16 (&dw)[NDW], TVA tA, int colA, TVB tB, int colB, int lane) {
  constexpr int KSTEPS = 4;
  h16x8 a[2][NOB], b[2][NIB];
#pragma unroll
  for (int o = 0; o < NOB; ++o) a[0][o] = tr_frag_s8(tA, colA + 32 * o, 0, lane);
#pragma unroll
  for (int i = 0; i < NIB; ++i) b[0][i] = tr_frag_s8(tB, colB + 32 * i, 0, lane);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    if (ks + 1 < KSTEPS) {
#pragma unroll
      for (int o = 0; o < NOB; ++o) a[(ks + 1) & 1][o] = tr_frag_s8(tA, colA + 32 * o, ks + 1, lane);
#pragma unroll
      for (int i = 0; i < NIB; ++i) b[(ks + 1) & 1][i] = tr_frag_s8(tB, colB + 32 * i, ks + 1, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int o = 0; o < NOB; ++o)
#pragma unroll
      for (int i = 0; i < NIB; ++i) mfma_acc_a(dw[OFF + o * NIB + i], a[ks & 1][o], b[ks & 1][i]);
  }
}
#ifdef ALN_PHASE_TIMING
__device__ long long g_sp_cycles[4][8];
extern "C" int aln_debug_read_pair(long long* host_out, int reset) {
  if (reset) { long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_cycles), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sp_cycles), sizeof(long long) * 32);
}
#define SP_DECL long long sp_acc[8] = {0}, sp_last = clock64();
#define SP_STAMP(i) { long long sp_now = clock64(); sp_acc[i] += sp_now - sp_last; sp_last = sp_now; }
#define SP_FLUSH if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 8; ++i) g_sp_cycles[wave][i] += sp_acc[i];
#else
#define SP_DECL
#define SP_STAMP(i)
#define SP_FLUSH
#endif
template <int CP, bool DOTS>   // CP: padded class count, semantic_out's out_pad (16 or 32); DOTS: also <logits, g_sem> + <f, g_feat> per row
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_sem_bwd_pair(const h16* __restrict__ wf_f, const h16* __restrict__ wb_f, const h16* __restrict__ wf_o, const h16* __restrict__ wb_o,
                    const h16* __restrict__ sigma_out, const float* __restrict__ w_row, RowSrc rs, const float* __restrict__ g_sem, int C,
                    const float* __restrict__ g_feat, int G, int rows, h16* __restrict__ d_geo, float* __restrict__ ws_f,
                    float* __restrict__ ws_o, float* __restrict__ dots_row, int* __restrict__ found_inf) {
  constexpr int D = 64, KS = 4, KSO = CP / 16, PH = hid_pitch(64), PFG = 80 + 8, PDL = 32, TR = 64;
  constexpr int SET = TR * (PFG + 7 * PH + PDL + 16);   // halves per tile set
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = (rows + TR - 1) / TR;
  const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // tiles blockIdx.x, + gridDim.x, ...
  for (int i = threadIdx.x; i < 2 * SET / 8 + 1; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);   // (dL/dlogits columns >= CP stay zero; + the flag word)
  struct Tiles { PlainV<lds_h16*> FG, H1, H2, G1, DG1, DF, DH2, DH1, DL, DGO; };
  auto tiles_of = [&](int set) {
    lds_h16* t = (lds_h16*)smem + set * SET;
    Tiles T;
    T.FG = {t, PFG}; t += TR * PFG;      // relu(f) | geo_feat, 1   (semantic_out's input)
    T.H1 = {t, PH}; t += TR * PH;        // h1, h2 of semantic_features, g1 of semantic_out
    T.H2 = {t, PH}; t += TR * PH;
    T.G1 = {t, PH}; t += TR * PH;
    T.DG1 = {t, PH}; t += TR * PH;       // dL/d(pre-activation) of g1, f, h2, h1
    T.DF = {t, PH}; t += TR * PH;
    T.DH2 = {t, PH}; t += TR * PH;
    T.DH1 = {t, PH}; t += TR * PH;
    T.DL = {t, PDL}; t += TR * PDL;      // dL/dlogits (columns >= CP zero)
    T.DGO = {t, 16};                     // d(geo_feat, 1) of the semantic_out branch
    return T;
  };
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();   // zero fill done
  if (wave < 2) {
    // =================================================================================================== the chain
    const h16x8* const ff = (const h16x8*)wf_f;
    const h16x8* const of = (const h16x8*)wf_o; const h16x8* const ob = (const h16x8*)wb_o;
    h16x8 Wf0[2], Wf1[2][KS], Wf2[2][KS], Vf0[2][5], V1T[2][KSO], V0T[3][KS], Vf1[DOTS ? KS : 1];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      Wf0[m] = ff[(size_t)m * 64 + lane];                                                    // semantic_features: layer 0 (natural k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        Wf1[m][ks] = ff[(size_t)(2 + m * KS + ks) * 64 + lane];                              //   layer 1 (chained k)
        Wf2[m][ks] = ff[(size_t)(2 + 2 * KS + m * KS + ks) * 64 + lane];                     //   last layer (chained k)
      }
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) Vf0[m][ks] = of[(size_t)(m * 5 + ks) * 64 + lane];      // semantic_out: layer 0 (natural k over relu(f) | geo)
#pragma unroll
      for (int ks = 0; ks < KSO; ++ks) V1T[m][ks] = ob[(size_t)(m * KSO + ks) * 64 + lane];  //   backward: V1^T (natural o over the logits)
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) V0T[m][ks] = ob[(size_t)(2 * KSO + m * KS + ks) * 64 + lane];   // semantic_out: V0^T (80 inputs, chained o)
    if constexpr (DOTS) {   // semantic_out's last layer, forward: the logits are recomputed for their dot product with g_sem
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) Vf1[ks] = of[(size_t)(2 * 5 + ks) * 64 + lane];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the loads have landed before the loop is entered (mlp_bwd128.hip: hipcc's wait-count pass)
    const h16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int srow = 32 * wave + c;
    // One wave per SIMD hides no latency by itself: the row inputs of tile it + 1 (geo_feat chunk, ray, weight) are fetched while
    // tile it is in the chain, and the per-ray output gradients of tile it are requested before its first MFMA (they are needed
    // after 26 and 50 of them).  Without this the four dependent trips to L2/HBM were 60 % of the chain's time (measured).
    // (all of these loads are unconditional, from clamped rows: a branch around a load makes hipcc's wait-count pass fall back to
    //  vmcnt(0) at the join, which waits for the prefetch of the NEXT tile as well)
    const int last_row = rows - 1;
    auto geo_of = [&](h16x8 lo, h16x8 hi) __attribute__((always_inline)) {   // geo_chunk() of the raw sigma_out row, without a divergent branch
      h16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const h16 v0 = (j < 7) ? lo[j + 1] : hi[0], v1 = (j < 7) ? hi[j + 1] : (h16)1.0f;
        const h16 v = hf ? v1 : v0;
        o[j] = (8 * hf + j < G) ? v : (h16)1.0f;
      }
      return o;
    };
    // hipcc sinks plain loads down to their first use (one exposed round trip each, measured: 8 x vmcnt(0) in a row), and a
    // scheduling barrier does not hold them (instruction selection orders loads before the barrier exists).  So these loads are
    // asm statements with the idle AGPR half as destination, and the waits are placed by hand: memory loads return in order, a wait
    // names the number of YOUNGER loads that may still be in flight, and takes the registers as operands so that no use moves above it.
    f32x4 lon, hin; float wrn;
#define SP_LOAD32(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=a"(dst) : "v"(ptr))
#define SP_LOAD128(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(ptr))
    {
      const int row = min((int)blockIdx.x * TR + srow, last_row);
      SP_LOAD128(lon, sigma_out + (size_t)row * 16); SP_LOAD128(hin, sigma_out + (size_t)row * 16 + 8); SP_LOAD32(wrn, w_row + row);
    }
    SP_DECL
    for (int it = 0; it <= my_tiles; ++it) {
      if (it < my_tiles) {
        const Tiles T = tiles_of(it & 1);
        const int row = ((int)blockIdx.x + it * (int)gridDim.x) * TR + srow;
        const bool valid = row < rows;
        f32x16 acc[2];
        h16x8 p1[4], p2[4], pd[4];
        const int ray = row_ray(rs, min(row, last_row));
        float gs[KSO][8], gsc[DOTS ? CP / 2 : 1];
        f32x4 gf[2][4];
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) SP_LOAD32(gs[ks][j], g_sem + (size_t)ray * C + min(16 * ks + 8 * hf + j, C - 1));
        if constexpr (DOTS) {   // g_sem once more, in the accumulator's row order: class (r & 3) + 8 (r >> 2) + 4 half of register r
#pragma unroll
          for (int r = 0; r < CP / 2; ++r) SP_LOAD32(gsc[r], g_sem + (size_t)ray * C + min((r & 3) + 8 * (r >> 2) + 4 * hf, C - 1));
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) SP_LOAD128(gf[m][q], g_feat + (size_t)ray * D + 32 * m + 8 * q + 4 * hf);
        asm volatile("s_waitcnt vmcnt(%3)" : "+a"(lon), "+a"(hin), "+a"(wrn) : "i"(8 * KSO + 8 + (DOTS ? CP / 2 : 0)));   // the loads above are younger
        const h16x8 x0 = valid ? geo_of(__builtin_bit_cast(h16x8, lon), __builtin_bit_cast(h16x8, hin)) : z8;
        const float wr = valid ? wrn : 0.f;   // (zero weight: the row contributes nothing to any gradient)
        *(LDS_VEC(h16x8)*)T.FG.at(srow, D + 8 * hf) = x0;
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = mfma16(Wf0[m], x0, zero16);
        relu_pack_store<2>(acc, p1, nullptr, hf);
        write_packed_tile<2>(T.H1, srow, p1, hf);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(Wf1[m][ks], p1[ks], ks == 0 ? zero16 : acc[m]);
        relu_pack_store<2>(acc, p2, nullptr, hf);
        write_packed_tile<2>(T.H2, srow, p2, hf);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(Wf2[m][ks], p2[ks], ks == 0 ? zero16 : acc[m]);
        // DOTS: <f_s, g_feat[ray]> + <logits_s, g_sem[ray]> for the wave's 32 rows through the matrix pipe: the per-ray gradient (fp16, like
        // every gradient of the chain) is the A operand -- the same fragment in every lane row, so every row of the product holds the dot
        // products -- and the packed fp16 activations are the B operand they already are for the next layer.  g_feat / g_sem sit in this
        // lane's registers in exactly the chained k-order of its half (gf[m][q]: features 32 m + 8 q + 4 half + 0..3).
        f32x16 dacc = zero16;
        if constexpr (DOTS) {
          asm volatile("s_waitcnt vmcnt(0)" : "+a"(gf[0][0]), "+a"(gf[0][1]), "+a"(gf[0][2]), "+a"(gf[0][3]), "+a"(gf[1][0]), "+a"(gf[1][1]),
                       "+a"(gf[1][2]), "+a"(gf[1][3]));
#pragma unroll
          for (int r = 0; r < CP / 2; r += 8)
            asm volatile("" : "+a"(gsc[r]), "+a"(gsc[r + 1]), "+a"(gsc[r + 2]), "+a"(gsc[r + 3]), "+a"(gsc[r + 4]), "+a"(gsc[r + 5]), "+a"(gsc[r + 6]),
                         "+a"(gsc[r + 7]));
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const f32x4 g0 = gf[ks >> 1][2 * (ks & 1)], g1 = gf[ks >> 1][2 * (ks & 1) + 1];
            union { uint32_t w[4]; h16x8 v; } ga, fb;
            ga.w[0] = cvt_pk(g0.x, g0.y); ga.w[1] = cvt_pk(g0.z, g0.w); ga.w[2] = cvt_pk(g1.x, g1.y); ga.w[3] = cvt_pk(g1.z, g1.w);
#pragma unroll
            for (int j = 0; j < 8; j += 2) fb.w[j >> 1] = cvt_pk(acc[ks >> 1][8 * (ks & 1) + j], acc[ks >> 1][8 * (ks & 1) + j + 1]);   // f as stored (fp16), before the ReLU
            dacc = mfma16(ga.v, fb.v, dacc);
          }
        }
        // f is rounded to fp16 first (what the forward stored), then relu'd: relu(f) feeds semantic_out and masks dL/df of that branch
        relu_pack_store<2>(acc, pd, nullptr, hf);
        write_packed_tile<2>(T.FG, srow, pd, hf);
        // dL/dlogits of the rows: w * g_sem[ray], natural order, also parked for dW(V1)
        h16x8 dl[KSO];
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks)   // the 8 loads of g_feat are younger
          asm volatile("s_waitcnt vmcnt(8)" : "+a"(gs[ks][0]), "+a"(gs[ks][1]), "+a"(gs[ks][2]), "+a"(gs[ks][3]), "+a"(gs[ks][4]), "+a"(gs[ks][5]),
                       "+a"(gs[ks][6]), "+a"(gs[ks][7]));
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks) {
#pragma unroll
          for (int j = 0; j < 8; ++j) dl[ks][j] = (h16)((16 * ks + 8 * hf + j < C) ? wr * gs[ks][j] : 0.f);
          *(LDS_VEC(h16x8)*)T.DL.at(srow, 16 * ks + 8 * hf) = dl[ks];
        }
        {   // the next tile's row inputs: requested here, after the wait for g_sem (memory loads return in order), used in ~5000 cycles
          const int rown = min(it + 1 < my_tiles ? row + (int)gridDim.x * TR : row, last_row);
          SP_LOAD128(lon, sigma_out + (size_t)rown * 16); SP_LOAD128(hin, sigma_out + (size_t)rown * 16 + 8); SP_LOAD32(wrn, w_row + rown);
        }
        // semantic_out layer 0 reads its input in natural k order: the wave's own rows back from the tile
        h16x8 xb[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) xb[ks] = *(const LDS_VEC(h16x8)*)T.FG.at(srow, 16 * ks + 8 * hf);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(Vf0[m][ks], xb[ks], ks == 0 ? zero16 : acc[m]);
        relu_pack_store<2>(acc, pd, nullptr, hf);
        write_packed_tile<2>(T.G1, srow, pd, hf);
        if constexpr (DOTS) {   // the logits (fp16, as the forward stored them) against g_sem[ray], into the same accumulator
          f32x16 lg = zero16;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) lg = mfma16(Vf1[ks], pd[ks], lg);
#pragma unroll
          for (int ks = 0; ks < KSO; ++ks) {
            union { uint32_t w[4]; h16x8 v; } ga, lb;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const int c0 = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);   // class of chained position j (and j + 1: c0 + 1)
              ga.w[j >> 1] = cvt_pk(c0 < C ? gsc[8 * ks + j] : 0.f, c0 + 1 < C ? gsc[8 * ks + j + 1] : 0.f);
              lb.w[j >> 1] = cvt_pk(lg[8 * ks + j], lg[8 * ks + j + 1]);
            }
            dacc = mfma16(ga.v, lb.v, dacc);
          }
          if (valid && hf == 0) dots_row[row] = dacc[0];
        }
        // ---- backward: g1 (its ReLU mask from the packed g1 still in registers)
#pragma unroll
        for (int ks = 0; ks < KSO; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(V1T[m][ks], dl[ks], ks == 0 ? zero16 : acc[m]);
        mask_pack_lds<2>(acc, pd, T.G1, srow, hf);
        write_packed_tile<2>(T.DG1, srow, pd, hf);
        // ---- d(relu(f) | geo): 64 + 16 outputs (the geo block first: its 16 values leave the accumulator at once)
        h16x4 dgo[2];   // d(geo_feat, 1) of the semantic_out branch: features 8 q + 4 half + 0..3, q < 2
        {
          f32x16 ag = zero16;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) ag = mfma16(V0T[2][ks], pd[ks], ag);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) dgo[q][r] = (h16)ag[4 * q + r];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma16(V0T[m][ks], pd[ks], ks == 0 ? zero16 : acc[m]);
        {   // dL/df = w * g_feat[ray] + relu'(f) * dL/df of the semantic_out branch (that part rounded to fp16 first, as the two-launch path did)
          h16x8 dfo[4];
          mask_pack_lds<2>(acc, dfo, T.FG, srow, hf);
          asm volatile("s_waitcnt vmcnt(3)" : "+a"(gf[0][0]), "+a"(gf[0][1]), "+a"(gf[0][2]), "+a"(gf[0][3]), "+a"(gf[1][0]), "+a"(gf[1][1]),
                       "+a"(gf[1][2]), "+a"(gf[1][3]));   // the next tile's three loads are younger
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 g4 = gf[m][q];
              const h16x8 ov = dfo[2 * m + (q >> 1)];
              const int e = 4 * (q & 1);
              pd[2 * m + (q >> 1)][e] = (h16)(wr * g4.x + (float)ov[e]); pd[2 * m + (q >> 1)][e + 1] = (h16)(wr * g4.y + (float)ov[e + 1]);
              pd[2 * m + (q >> 1)][e + 2] = (h16)(wr * g4.z + (float)ov[e + 2]); pd[2 * m + (q >> 1)][e + 3] = (h16)(wr * g4.w + (float)ov[e + 3]);
            }
        }
        write_packed_tile<2>(T.DF, srow, pd, hf);
#pragma unroll
        for (int q = 0; q < 2; ++q) *(LDS_VEC(h16x4)*)T.DGO.at(srow, 8 * q + 4 * hf) = dgo[q];
      }
      SP_STAMP(0) __syncthreads(); SP_STAMP(1)   // set it & 1 is complete, set (it - 1) & 1 has been consumed
    }
    SP_FLUSH
  } else {
    // ============================================================== backward of semantic_features, then the weight gradients
    // The chain ends at dL/df: its waves would otherwise run twice as long as these two (measured 6800 against 3400 cycles per tile).
    // Waves 2 and 3 take 32 rows each of the previous tile through W2^T, W1^T, W0^T (20 MFMAs, fragments resident), write dL/dh2,
    // dL/dh1 and the rows' d(geo_feat); wave 3 then raises a flag in LDS, and wave 2 waits for it between dW(W2), which needs
    // nothing of this, and dW(W1), dW(W0), which read all 64 rows.
    const h16x8* const fb = (const h16x8*)wb_f;
    h16x8 W2T[2][KS], W1T[2][KS], W0T[KS];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        W2T[m][ks] = fb[(size_t)(m * KS + ks) * 64 + lane];              // W2^T (natural o), W1^T (chained o)
        W1T[m][ks] = fb[(size_t)(2 * KS + m * KS + ks) * 64 + lane];
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) W0T[ks] = fb[(size_t)(4 * KS + ks) * 64 + lane];   // W0^T (16 inputs, chained o)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    LDS_VEC(int)* const flag = (LDS_VEC(int)*)((lds_h16*)smem + 2 * SET);
    const int srow = 32 * (wave - 2) + c;
    h16x2 nanz = {0, 0};
    constexpr int NDW = 10;
    f32x16 dw[NDW];
#pragma unroll
    for (int b = 0; b < NDW; ++b) dw[b] = zero16;
    SP_DECL
    for (int it = 0; it <= my_tiles; ++it) {
      if (it > 0) {
        const Tiles T = tiles_of((it - 1) & 1);
        {
          const int row = ((int)blockIdx.x + (it - 1) * (int)gridDim.x) * TR + srow;
          f32x16 acc[2];
          h16x8 xb[KS], pd[4];
          // W2^T takes dL/df in natural order: the rows as the chain stored them
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) xb[ks] = *(const LDS_VEC(h16x8)*)T.DF.at(srow, 16 * ks + 8 * hf);
          h16x4 dgo[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) dgo[q] = *(const LDS_VEC(h16x4)*)T.DGO.at(srow, 8 * q + 4 * hf);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = mfma16(W2T[m][ks], xb[ks], ks == 0 ? zero16 : acc[m]);
          mask_pack_lds<2>(acc, pd, T.H2, srow, hf);
          write_packed_tile<2>(T.DH2, srow, pd, hf);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = mfma16(W1T[m][ks], pd[ks], ks == 0 ? zero16 : acc[m]);
          mask_pack_lds<2>(acc, pd, T.H1, srow, hf);
          write_packed_tile<2>(T.DH1, srow, pd, hf);
          f32x16 o = zero16;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) o = mfma16(W0T[ks], pd[ks], o);
          if (row < rows) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              h16x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (h16)(o[4 * q + r] + (float)dgo[q][r]);
              nanz = nan_fold((h16x2){v[0], v[1]}, nan_fold((h16x2){v[2], v[3]}, nanz));
              *(h16x4*)(d_geo + (size_t)row * 16 + 8 * q + 4 * hf) = v;
            }
          }
        }
        SP_STAMP(2)
        if (wave == 2) {
          dw_matrix<2, 2, 0>(dw, T.DF, 0, T.H2, 0, lane);      // W2 [64][64]: dL/df^T h2 (needs nothing of the other wave)
          SP_STAMP(3)
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          SP_STAMP(4)
          dw_matrix<2, 2, 4>(dw, T.DH2, 0, T.H1, 0, lane);     // W1 [64][64]
          dw_matrix<2, 1, 8>(dw, T.DH1, 0, T.FG, D, lane);     // W0 [64][16]: dL/dh1^T geo
        } else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // rows 32..63 of dL/dh2, dL/dh1 are in LDS
          __hip_atomic_store(flag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          dw_matrix<2, 3, 0>(dw, T.DG1, 0, T.FG, 0, lane);     // V0 [64][80]: dL/dg1^T [relu(f) | geo]  (columns 80..95: never flushed)
          dw_matrix<1, 2, 6>(dw, T.DL, 0, T.G1, 0, lane);      // V1 [CP][64]
        }
      }
      SP_STAMP(0) __syncthreads(); SP_STAMP(1)
    }
    SP_FLUSH
    // hipcc does not know that the asm statements above are MFMAs: the flush must wait for the last matrix pass to land
    asm volatile("s_nop 15" : "+a"(dw[0]), "+a"(dw[1]), "+a"(dw[2]), "+a"(dw[3]), "+a"(dw[4]));
    asm volatile("s_nop 1" : "+a"(dw[5]), "+a"(dw[6]), "+a"(dw[7]), "+a"(dw[8]), "+a"(dw[9]));
    // ---- slabs of partial sums (fixed-order reduction by k_dw_reduce_all), layout = each head's fp32 master block
    bool bad = false;
    auto flush = [&](f32x16& a, int ob_, int ib_, int OUTL, int INL, float* dst) {
      const int i = 32 * ib_ + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * ob_ + (r & 3) + 8 * (r >> 2) + 4 * hf;
        const bool in_w = o < OUTL && i < INL;
        bad |= in_w && !(fabsf(a[r]) <= 3.0e38f);
        if (in_w) dst[(size_t)o * INL + i] = a[r];
      }
    };
    if (ws_f && ws_o) {
      constexpr int NF = 16 * 64 + 2 * 64 * 64, NO = 80 * 64 + 64 * CP;
      float* const sf = ws_f + (size_t)blockIdx.x * NF;
      float* const so = ws_o + (size_t)blockIdx.x * NO;
      if (wave == 2) {
#pragma unroll
        for (int b = 0; b < 4; ++b) flush(dw[b], b >> 1, b & 1, 64, 64, sf + 16 * 64 + 64 * 64);
#pragma unroll
        for (int b = 0; b < 4; ++b) flush(dw[4 + b], b >> 1, b & 1, 64, 64, sf + 16 * 64);
#pragma unroll
        for (int b = 0; b < 2; ++b) flush(dw[8 + b], b, 0, 64, 16, sf);
      } else {
#pragma unroll
        for (int b = 0; b < 6; ++b) flush(dw[b], b / 3, b % 3, 64, 80, so);
#pragma unroll
        for (int b = 0; b < 2; ++b) flush(dw[6 + b], 0, b, CP, 64, so + 80 * 64);
      }
    }
    if (found_inf && __any(bad || nan_bad(nanz)) && lane == 0) atomicOr(found_inf, 1);
  }
}
static bool sem_pair_fused_ok(const AlnMlpDesc* semf, const AlnMlpDesc* semo, int D, int G) {
  return semf->in_pad == 16 && semf->hidden == 64 && semf->n_hidden == 2 && semf->out_pad == 64 && D == 64 && semo->in_pad == 80 &&
         semo->hidden == 64 && semo->n_hidden == 1 && semo->out_pad <= 32 /* OBO = 1: five weight-gradient blocks per wave */ && G <= 15 && semf->wf && semf->wb && semo->wf && semo->wb;
}
static int sem_pair_blocks(int rows) { const int t = (rows + 63) / 64; return t < 256 ? t : 256; }
// slabs per head the fused pair backward leaves in semf->dw_ws / semo->dw_ws for `rows` rows, or 0 when the pair runs as two launches
// (then aln_mlp_bwd_blocks of each head applies)
extern "C" int32_t aln_sem_heads_bwd_slabs(const AlnMlpDesc* semf, const AlnMlpDesc* semo, int32_t rows, int32_t D, int32_t G) {
  return (semf && semo && rows > 0 && sem_pair_fused_ok(semf, semo, D, G)) ? sem_pair_blocks(rows) : 0;
}

// ---------------------------------------------------------------- launchers
static int mlp_grid(int rows) {
  // blocks per launch of the forward / data-gradient kernels (256 threads, weights in LDS): 3 per CU
  // (the 128-wide heads' fragment images are 48-52 KB and the kernels use <= 168 VGPRs, so three blocks fit a CU)
  const int cap = 768;
  int tiles = (rows + 127) / 128;
  int g = tiles < cap ? tiles : cap;
  return g < 1 ? 1 : g;
}

static int mlp_fwd_src(const AlnMlpDesc* m, RowSrc xs, int32_t rows, const int32_t* rows_dev, void* h1, void* h2, void* out,
                       void* stream, float* sigma = nullptr) {
  ALN_REQUIRE(m && (xs.a || xs.mode == SRC_COLOR_IN) && out && m->wf, "mlp_fwd: NULL pointer");
  ALN_REQUIRE(m->in_pad % 16 == 0 && m->out_pad % 16 == 0, "mlp_fwd: widths must be multiples of 16");
  ALN_REQUIRE(m->in_pad <= 80, "mlp_fwd: in_pad %d exceeds the 80 input features the fused kernel holds in registers", m->in_pad);
  if (rows <= 0) return 0;
#ifndef ALN_FWD128_OLD   // (dev builds only: A/B against k_mlp_fwd)
  if (xs.mode == SRC_PLAIN && !h1 && !h2 && xs.lda == m->in_pad && ((uintptr_t)xs.a & 15) == 0) {   // the 128-wide heads over plain rows: mlp_fwd128.hip
    const int rc = aln_launch_fwd128(m, xs.a, rows, rows_dev, out, sigma, (hipStream_t)stream);
    if (rc != -3) { ALN_CHECK_LAUNCH("mlp_fwd128"); return rc; }
  }
#endif
  ALN_REQUIRE(!m->x_tiled, "mlp_fwd: tiled input rows (AlnMlpDesc.x_tiled) are read by the 128-wide kernel only");
  size_t halves = (size_t)aln_mlp_frag_halves(m->in_pad, m->hidden, m->out_pad, m->n_hidden, 0);
  size_t lds = halves * 2;
  ALN_REQUIRE(lds <= 160 * 1024, "mlp_fwd: weights (%zu B) exceed LDS", lds);
  dim3 g(mlp_grid(rows)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_KP(H, N, K, P)                                                                                          \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)k_mlp_fwd<H, N, K, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    hipLaunchKernelGGL((k_mlp_fwd<H, N, K, P>), g, b, lds, s, (const h16*)m->wf, halves, m->in_pad, m->out_pad, xs,    \
                       rows, rows_dev, (h16*)h1, (h16*)h2, (h16*)out, sigma);                                          \
  } while (0)
#define LAUNCH_K(H, N, K)                                                                                              \
  do {                                                                                                                 \
    if (xs.mode == SRC_PLAIN) LAUNCH_KP(H, N, K, true); else LAUNCH_KP(H, N, K, false);                                \
  } while (0)
#define LAUNCH(H, N)                                                                                                   \
  do {                                                                                                                 \
    switch (m->in_pad / 16) {                                                                                          \
      case 1: LAUNCH_K(H, N, 1); break;                                                                                \
      case 2: LAUNCH_K(H, N, 2); break;                                                                                \
      case 3: LAUNCH_K(H, N, 3); break;                                                                                \
      case 4: LAUNCH_K(H, N, 4); break;                                                                                \
      default: LAUNCH_K(H, N, 5); break;                                                                               \
    }                                                                                                                  \
  } while (0)
  if (m->hidden == 128 && m->n_hidden == 2) LAUNCH(128, 2);
  else if (m->hidden == 128 && m->n_hidden == 1) LAUNCH(128, 1);
  else if (m->hidden == 64 && m->n_hidden == 2) LAUNCH(64, 2);
  else if (m->hidden == 64 && m->n_hidden == 1) LAUNCH(64, 1);
  else { aln_set_error("mlp_fwd: unsupported hidden=%d n_hidden=%d", m->hidden, m->n_hidden); return -1; }
#undef LAUNCH
#undef LAUNCH_K
#undef LAUNCH_KP
  ALN_CHECK_LAUNCH("mlp_fwd");
  return 0;
}

extern "C" int aln_mlp_fwd(const AlnMlpDesc* m, const void* x, int32_t rows, const int32_t* rows_dev, void* h1, void* h2,
                           void* out, void* stream) {
  ALN_REQUIRE(m && x, "mlp_fwd: NULL pointer");
  return mlp_fwd_src(m, plain_src(x, m->in_pad), rows, rows_dev, h1, h2, out, stream);
}
// the density head: the same launch also writes sigma[row] = exp(out[row][0]) (ALNetwork.density, autolabel/models.py:175-188:
// sigma = trunc_exp(h[..., 0]); aln_sigma_act as an epilogue)
extern "C" int aln_density_fwd(const AlnMlpDesc* m, const void* x, int32_t rows, void* h1, void* h2, void* out, float* sigma,
                               void* stream) {
  ALN_REQUIRE(m && x && sigma, "density_fwd: NULL pointer");
  return mlp_fwd_src(m, plain_src(x, m->in_pad), rows, nullptr, h1, h2, out, stream, sigma);
}

// bytes of AlnMlpDesc.dw_ws: one fp32 slab of all the head's weights per backward block (at most 512 blocks)
extern "C" int64_t aln_mlp_dw_ws_bytes(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden) {
  const MlpLayers L = mlp_layers(in_pad, hidden, out_pad, n_hidden);
  return (int64_t)512 * (int64_t)(L.w_off[L.n - 1] + (size_t)L.in_[L.n - 1] * L.out_[L.n - 1]) * (int64_t)sizeof(float);
}

// heads whose forward AND backward kernels read tiled input rows (AlnMlpDesc.x_tiled): the 128-wide kernels of mlp_fwd128.hip / mlp_bwd128.hip
extern "C" int aln_mlp_supports_tiled(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden) {
  return hidden == 128 && n_hidden == 2 && out_pad == 16 && (in_pad == 32 || in_pad == 48);
}
// shapes the recompute backward is instantiated for (the list of TRYR below)
extern "C" int aln_mlp_has_recompute(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden) {
  static const int shapes[][4] = {{48, 128, 16, 2}, {32, 128, 16, 2}, {64, 128, 16, 2}, {16, 64, 64, 2},
                                  {80, 64, 16, 1},  {80, 64, 32, 1},  {80, 64, 48, 1},  {80, 64, 64, 1}};
  for (auto& q : shapes)
    if (q[0] == in_pad && q[1] == hidden && q[2] == out_pad && q[3] == n_hidden) return 1;
  return 0;
}

static int mlp_bwd_recomp_src(const AlnMlpDesc* m, RowSrc xs, RowSrc ds, int rows, const int* rows_dev, void* d_in, float* dW,
                              int* found_inf, hipStream_t s) {
  ALN_REQUIRE(m && m->wr && xs.a && (ds.a || ds.g), "mlp_bwd: recompute path needs x, dL/dout and the row-major weight copy (wr)");
  if (rows <= 0) return 0;
#ifndef ALN_BWD128_OLD   // (dev builds only: scripts/dev/build_variant.py A/B against round 3's kernel)
  if (m->hidden == 128 && m->n_hidden == 2 && m->out_pad == 16 && xs.mode == SRC_PLAIN && ds.mode == SRC_PLAIN && m->wf && m->wb &&
      xs.lda == m->in_pad && ds.lda == m->out_pad) {   // the 128-wide heads: feature-sliced kernel of mlp_bwd128.hip
    const int g = bwd_recomp_blocks(m, rows);
    const MlpLayers LL = mlp_layers(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
    const int n_w = (int)(LL.w_off[LL.n - 1] + (size_t)LL.in_[LL.n - 1] * LL.out_[LL.n - 1]);
    ALN_REQUIRE(!dW || (m->dw_ws && (size_t)m->dw_ws_bytes >= (size_t)g * n_w * sizeof(float)),
                "mlp_bwd: AlnMlpDesc.dw_ws must hold %d slabs of %d floats (aln_mlp_dw_ws_bytes)", g, n_w);
    float* ws = dW ? (float*)m->dw_ws : nullptr;
    const int rc = aln_launch_bwd128(m, xs.a, ds.a, rows, rows_dev, d_in, ws, g, found_inf, s);
    ALN_REQUIRE(rc != -3 || !m->x_tiled, "mlp_bwd: tiled input rows need the 128-wide kernel");
    if (rc != -3) {
      ALN_CHECK_LAUNCH("mlp_bwd128");
      if (ws && !m->defer_dw_reduce) {
        hipLaunchKernelGGL(k_dw_reduce, dim3((n_w + DWR_E - 1) / DWR_E), dim3(DWR_G * DWR_E), 0, s, ws, g, n_w, dW);
        ALN_CHECK_LAUNCH("dw_reduce");
      }
      return 0;
    }
  }
#endif
  ALN_REQUIRE(!m->x_tiled, "mlp_bwd: tiled input rows (AlnMlpDesc.x_tiled) are read by the 128-wide kernel only");
#define TRYR(I, H, O, N)                                                                                  \
  if (m->in_pad == I && m->hidden == H && m->out_pad == O && m->n_hidden == N)                            \
    return launch_bwd_recomp<I, H, O, N>(m, xs, ds, rows, rows_dev, d_in, dW, found_inf, s);
  TRYR(48, 128, 16, 2) TRYR(32, 128, 16, 2) TRYR(64, 128, 16, 2) TRYR(16, 64, 64, 2) TRYR(80, 64, 16, 1) TRYR(80, 64, 32, 1) TRYR(80, 64, 48, 1) TRYR(80, 64, 64, 1)
#undef TRYR
  aln_set_error("mlp_bwd: no recompute kernel for in=%d hid=%d out=%d nh=%d (pass saved activations)", m->in_pad, m->hidden,
                m->out_pad, m->n_hidden);
  return -1;
}

static int launch_dw(const h16* dA, int OW, const h16* X, int IW, int rows, const int* rows_dev, float* dW, hipStream_t s) {
  size_t lds = (size_t)DW_ROWS * (ceil32(OW) * 32 + ceil32(IW) * 32) * 2;
  ALN_REQUIRE(ceil32(OW) * ceil32(IW) <= 16, "dw_gemm: %dx%d exceeds 16 C-blocks", OW, IW);
  int chunks = (rows + DW_ROWS - 1) / DW_ROWS;
  int g = chunks < 256 ? chunks : 256;
  hipFuncSetAttribute((const void*)k_dw_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_dw_gemm, dim3(g), dim3(256), lds, s, dA, OW, X, IW, rows, rows_dev, dW);
  ALN_CHECK_LAUNCH("dw_gemm");
  return 0;
}

// The density head's recompute backward with its dL/dout rows ASSEMBLED BY THE KERNEL'S LOADER from their three producers (round 6: no
// aln_assemble_grads pass, no d_sigma_out buffer): row r = [ d_h0[r] | d_semf_in[r][0..G) + d_color_in[cidx_row[r]][16 .. 16 + G) ], the
// colour term only where cidx_row[r] >= 0 -- the arithmetic of aln_assemble_grads with d_semo_in = NULL (the fused semantic pair has folded
// its skip connection into d_semf_in).  128-wide two-hidden-layer head with 48 inputs and 16 outputs only (returns -3 otherwise).
extern "C" int aln_mlp_bwd_dso(const AlnMlpDesc* m, const void* x, const float* d_h0, const void* d_semf_in, const void* d_color_in,
                               const int32_t* cidx_row, int32_t G, int32_t rows, void* d_in, float* dW, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(m && x && d_h0 && d_semf_in && d_color_in && cidx_row && m->wb && m->wf, "mlp_bwd_dso: NULL pointer");
  ALN_REQUIRE(G >= 0 && G <= 15, "mlp_bwd_dso: G = %d out of range", G);
  if (m->hidden != 128 || m->n_hidden != 2 || m->out_pad != 16 || m->in_pad != 48) return -3;
  if (rows <= 0) return 0;
  const int g = bwd_recomp_blocks(m, rows);
  const MlpLayers LL = mlp_layers(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
  const int n_w = (int)(LL.w_off[LL.n - 1] + (size_t)LL.in_[LL.n - 1] * LL.out_[LL.n - 1]);
  ALN_REQUIRE(!dW || (m->dw_ws && (size_t)m->dw_ws_bytes >= (size_t)g * n_w * sizeof(float)),
              "mlp_bwd_dso: AlnMlpDesc.dw_ws must hold %d slabs of %d floats (aln_mlp_dw_ws_bytes)", g, n_w);
  float* ws = dW ? (float*)m->dw_ws : nullptr;
  const AlnDsoSrc dso{d_h0, d_semf_in, d_color_in, cidx_row, G};
  const int rc = aln_launch_bwd128(m, x, nullptr, rows, nullptr, d_in, ws, g, found_inf, (hipStream_t)stream, &dso);
  if (rc) return rc;
  ALN_CHECK_LAUNCH("mlp_bwd128_dso");
  if (ws && !m->defer_dw_reduce) {
    hipLaunchKernelGGL(k_dw_reduce, dim3((n_w + DWR_E - 1) / DWR_E), dim3(DWR_G * DWR_E), 0, (hipStream_t)stream, ws, g, n_w, dW);
    ALN_CHECK_LAUNCH("dw_reduce");
  }
  return 0;
}

extern "C" int aln_mlp_bwd(const AlnMlpDesc* m, const void* x, const void* h1, const void* h2, const void* d_out,
                           int32_t rows, const int32_t* rows_dev, void* dA1, void* dA2, void* d_in, float* dW,
                           int32_t* found_inf, void* stream) {
  ALN_REQUIRE(m && d_out && m->wb, "mlp_bwd: NULL pointer");
  if (!h1) {  // no saved activations: recompute them inside the fused kernel
    ALN_REQUIRE(x, "mlp_bwd: recompute path needs x");
    return mlp_bwd_recomp_src(m, plain_src(x, m->in_pad), plain_src(d_out, m->out_pad), rows, rows_dev, d_in, dW, found_inf,
                              (hipStream_t)stream);
  }
  ALN_REQUIRE(m->n_hidden == 1 || h2, "mlp_bwd: h2 required for 2 hidden layers");
  ALN_REQUIRE(!dW || (x && dA1 && (m->n_hidden == 1 || dA2)), "mlp_bwd: dW needs x, dA1, dA2 buffers");
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  size_t halves = (size_t)aln_mlp_frag_halves(m->in_pad, m->hidden, m->out_pad, m->n_hidden, 1);
  size_t lds = halves * 2;
  ALN_REQUIRE(lds <= 160 * 1024, "mlp_bwd: weights (%zu B) exceed LDS", lds);
  dim3 g(mlp_grid(rows)), b(256);
#define LAUNCH(H, N)                                                                                                   \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)k_mlp_bwd<H, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL((k_mlp_bwd<H, N>), g, b, lds, s, (const h16*)m->wb, halves, m->in_pad, m->out_pad, (const h16*)h1, \
                       (const h16*)h2, (const h16*)d_out, rows, rows_dev, (h16*)dA1, (h16*)dA2, (h16*)d_in, found_inf); \
  } while (0)
  if (m->hidden == 128 && m->n_hidden == 2) LAUNCH(128, 2);
  else if (m->hidden == 128 && m->n_hidden == 1) LAUNCH(128, 1);
  else if (m->hidden == 64 && m->n_hidden == 2) LAUNCH(64, 2);
  else if (m->hidden == 64 && m->n_hidden == 1) LAUNCH(64, 1);
  else { aln_set_error("mlp_bwd: unsupported hidden=%d n_hidden=%d", m->hidden, m->n_hidden); return -1; }
#undef LAUNCH
  ALN_CHECK_LAUNCH("mlp_bwd");
  if (dW) {
    MlpLayers L = mlp_layers(m->in_pad, m->hidden, m->out_pad, m->n_hidden);
    int rc;
    // layer 0: dA1 x X
    if ((rc = launch_dw((const h16*)dA1, m->hidden, (const h16*)x, m->in_pad, rows, rows_dev, dW + L.w_off[0], s))) return rc;
    if (m->n_hidden == 2) {
      if ((rc = launch_dw((const h16*)dA2, m->hidden, (const h16*)h1, m->hidden, rows, rows_dev, dW + L.w_off[1], s))) return rc;
      if ((rc = launch_dw((const h16*)d_out, m->out_pad, (const h16*)h2, m->hidden, rows, rows_dev, dW + L.w_off[2], s))) return rc;
    } else {
      if ((rc = launch_dw((const h16*)d_out, m->out_pad, (const h16*)h1, m->hidden, rows, rows_dev, dW + L.w_off[1], s))) return rc;
    }
  }
  return 0;
}

// color_net on the live samples with its input rows built inside the kernel (inference: nothing to keep for a backward)
extern "C" int aln_color_fwd(const AlnMlpDesc* color, const int32_t* live_idx, const int32_t* n_live, int32_t max_rows,
                             const float* rays_d, const float* dirs, int32_t N, int32_t S1, int32_t S2, const void* sigma_out,
                             int32_t G, void* color_out, void* stream) {
  ALN_REQUIRE(color && (rays_d || dirs) && sigma_out && color_out && (!live_idx || n_live), "color_fwd: NULL pointer");
  ALN_REQUIRE(color->in_pad == 32 && G + 1 <= 16, "color_fwd: needs in_pad 32 (SH16 + geo_feat <= 15), got in_pad %d G %d", color->in_pad, G);
  RowSrc x{}; x.mode = SRC_COLOR_IN; x.b = (const h16*)sigma_out; x.ldb = 16; x.G = G; x.idx = live_idx;
  x.g = dirs ? dirs : rays_d; x.gw = dirs ? 1 : 0; row_src_rays(x, N, S1, S2);
  return mlp_fwd_src(color, x, max_rows, live_idx ? n_live : nullptr, nullptr, nullptr, color_out, stream);
}

// ---------------------------------------------------------------- semantic heads with on-the-fly inputs / gradients
// semantic_features(geo_feat) -> f ; semantic_out(cat[relu(f), geo_feat]) -> logits   (autolabel/models.py:248-256)
extern "C" int aln_sem_heads_fwd(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, int32_t rows, int32_t D,
                                 int32_t G, void* feat, void* logits, void* stream) {
  ALN_REQUIRE(semf && semo && sigma_out && feat && logits, "sem_heads_fwd: NULL pointer");
  ALN_REQUIRE(semf->out_pad == D && semo->in_pad >= D + G + 1 && D % 8 == 0, "sem_heads_fwd: shape mismatch");
  if (semf->in_pad == 16 && semf->hidden == 64 && semf->n_hidden == 2 && semf->out_pad == 64 && D == 64 && semo->in_pad == 80 &&
      semo->hidden == 64 && semo->n_hidden == 1 && semo->out_pad <= 64 && semf->wf && semo->wf) {
    if (rows <= 0) return 0;
    size_t hf_ = (size_t)aln_mlp_frag_halves(16, 64, 64, 2, 0), ho_ = (size_t)aln_mlp_frag_halves(80, 64, semo->out_pad, 1, 0);
    size_t lds = (((hf_ + 7) & ~(size_t)7) + ((ho_ + 7) & ~(size_t)7) + 4 * 32 * (64 + 8)) * 2;
    hipFuncSetAttribute((const void*)k_sem_fwd_fused<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_sem_fwd_fused<1, false>), dim3(mlp_grid(rows)), dim3(256), lds, (hipStream_t)stream, (const h16*)semf->wf, hf_,
                       (const h16*)semo->wf, ho_, (const h16*)sigma_out, rows, G, semo->out_pad, (h16*)feat, (h16*)logits, nullptr, nullptr);
    ALN_CHECK_LAUNCH("sem_fwd_fused");
    return 0;
  }
  RowSrc a{}; a.mode = SRC_SEMF_IN; a.a = (const h16*)sigma_out; a.lda = 16; a.G = G; a.D = D;
  if (int rc = mlp_fwd_src(semf, a, rows, nullptr, nullptr, nullptr, feat, stream)) return rc;
  RowSrc b{}; b.mode = SRC_SEMO_IN; b.a = (const h16*)feat; b.lda = D; b.b = (const h16*)sigma_out; b.ldb = 16; b.G = G; b.D = D;
  return mlp_fwd_src(semo, b, rows, nullptr, nullptr, nullptr, logits, stream);
}

// The training step's forward of both heads: per-tile weighted sums instead of the rows (k_sem_fwd_fused<SUMS>).
// tile_sums: [ceil(rows / 32)][96] fp32 -- columns [0, 64) = sum over the tile's rows of w_row * f, [64, 64 + out_pad) the same of
// the logits.  The caller guarantees that a 32-row tile never straddles two rays (S1 % 32 == 0, S2 % 32 == 0).
extern "C" int aln_sem_heads_fwd_sums(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, int32_t rows, int32_t D,
                                      int32_t G, const float* w_row, float* tile_sums, void* stream) {
  ALN_REQUIRE(semf && semo && sigma_out && w_row && tile_sums, "sem_heads_fwd_sums: NULL pointer");
  ALN_REQUIRE(sem_pair_fused_ok(semf, semo, D, G) && semo->out_pad <= 32 && semo->out_pad % 8 == 0 && semf->wf && semo->wf,
              "sem_heads_fwd_sums: both heads 64 wide, D = 64, <= 32 padded classes (ask aln_sem_heads_bwd_slabs)");
  ALN_REQUIRE(rows % 32 == 0, "sem_heads_fwd_sums: whole 32-row tiles only (%d rows)", rows);
  if (rows <= 0) return 0;
  size_t hf_ = (size_t)aln_mlp_frag_halves(16, 64, 64, 2, 0), ho_ = (size_t)aln_mlp_frag_halves(80, 64, semo->out_pad, 1, 0);
  size_t lds = (((hf_ + 7) & ~(size_t)7) + ((ho_ + 7) & ~(size_t)7) + 4 * 32 * (64 + 8)) * 2;
  hipFuncSetAttribute((const void*)k_sem_fwd_fused<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_sem_fwd_fused<1, true>), dim3(mlp_grid(rows)), dim3(256), lds, (hipStream_t)stream, (const h16*)semf->wf, hf_,
                     (const h16*)semo->wf, ho_, (const h16*)sigma_out, rows, G, semo->out_pad, nullptr, nullptr, w_row, tile_sums);
  ALN_CHECK_LAUNCH("sem_fwd_sums");
  return 0;
}

// backward of both heads from the per-ray output gradients: dL/dlogits[row] = w_row * g_sem[ray],
// dL/df[row] = w_row * g_feat[ray] + relu'(f) * dL/d(semo input)[row][:D].  Writes d_semo_in [rows, semo.in_pad] and
// d_semf_in [rows, semf.in_pad] (their geo_feat columns feed the sigma head) and accumulates both heads' dW.
extern "C" int aln_sem_heads_bwd(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, const void* feat,
                                 const float* w_row, const float* g_sem, const float* g_feat, int32_t N, int32_t S1, int32_t S2,
                                 int32_t C, int32_t rows, int32_t D, int32_t G, void* d_semo_in, void* d_semf_in, float* dW_semf,
                                 float* dW_semo, int32_t fold_geo, float* dots_row, int32_t* found_inf, void* stream) {
  const bool pair = fold_geo && dW_semf && dW_semo && semf && semo && sem_pair_fused_ok(semf, semo, D, G);
  ALN_REQUIRE(semf && semo && sigma_out && (feat || pair) && w_row && g_sem && g_feat && (d_semo_in || pair) && d_semf_in,
              "sem_heads_bwd: NULL pointer");
  ALN_REQUIRE(!dots_row || (pair && S1 % 32 == 0 && S2 % 32 == 0),
              "sem_heads_bwd: the per-row dot products need the one-kernel path (fold_geo, both weight gradients, D = 64) and 32-row blocks of ONE ray each "
              "(S1, S2 multiples of 32): the ray's gradient is one matrix operand for the block");
  ALN_REQUIRE(!fold_geo || (semf->in_pad == 16 && aln_mlp_has_recompute(semf->in_pad, semf->hidden, semf->out_pad, semf->n_hidden)),
              "sem_heads_bwd: fold_geo needs the 16-wide recompute backward of semantic_features");
  hipStream_t s = (hipStream_t)stream;
  if (pair) {   // the training step: one kernel for both heads
    if (rows <= 0) return 0;
    const int g = sem_pair_blocks(rows);
    const size_t nf = 16 * 64 + 2 * 64 * 64, no = 80 * 64 + 64 * (size_t)semo->out_pad;
    ALN_REQUIRE(semf->dw_ws && semo->dw_ws && (size_t)semf->dw_ws_bytes >= g * nf * sizeof(float) && (size_t)semo->dw_ws_bytes >= g * no * sizeof(float),
                "sem_heads_bwd: dw_ws of both heads must hold %d slabs", g);
    RowSrc rs{}; row_src_rays(rs, N, S1, S2);
    constexpr int PH = hid_pitch(64);
    const size_t lds = 2 * (size_t)64 * (88 + 7 * PH + 32 + 16) * 2 + 16;   // two sets of 64-row tiles, the flag word
#define LAUNCH_PAIR(CP, DOTS)                                                                                                        \
    do {                                                                                                                             \
      hipFuncSetAttribute((const void*)k_sem_bwd_pair<CP, DOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      hipLaunchKernelGGL((k_sem_bwd_pair<CP, DOTS>), dim3(g), dim3(256), lds, s, (const h16*)semf->wf, (const h16*)semf->wb,         \
                         (const h16*)semo->wf, (const h16*)semo->wb, (const h16*)sigma_out, w_row, rs, g_sem, C, g_feat, G, rows,     \
                         (h16*)d_semf_in, (float*)semf->dw_ws, (float*)semo->dw_ws, dots_row, found_inf);                            \
    } while (0)
    if (semo->out_pad == 16) { if (dots_row) LAUNCH_PAIR(16, true); else LAUNCH_PAIR(16, false); }
    else { if (dots_row) LAUNCH_PAIR(32, true); else LAUNCH_PAIR(32, false); }
#undef LAUNCH_PAIR
    ALN_CHECK_LAUNCH("sem_bwd_pair");
    if (!semf->defer_dw_reduce || !semo->defer_dw_reduce) {
      hipLaunchKernelGGL(k_dw_reduce, dim3(((int)nf + DWR_E - 1) / DWR_E), dim3(DWR_G * DWR_E), 0, s, (const float*)semf->dw_ws, g, (int)nf, dW_semf);
      hipLaunchKernelGGL(k_dw_reduce, dim3(((int)no + DWR_E - 1) / DWR_E), dim3(DWR_G * DWR_E), 0, s, (const float*)semo->dw_ws, g, (int)no, dW_semo);
      ALN_CHECK_LAUNCH("dw_reduce");
    }
    return 0;
  }
  RowSrc xo{}; xo.mode = SRC_SEMO_IN; xo.a = (const h16*)feat; xo.lda = D; xo.b = (const h16*)sigma_out; xo.ldb = 16; xo.G = G; xo.D = D;
  RowSrc go{}; go.mode = SRC_DLOGITS; go.w_row = w_row; go.g = g_sem; go.gw = C; row_src_rays(go, N, S1, S2);
  if (int rc = mlp_bwd_recomp_src(semo, xo, go, rows, nullptr, d_semo_in, dW_semo, found_inf, s)) return rc;
  RowSrc xf{}; xf.mode = SRC_SEMF_IN; xf.a = (const h16*)sigma_out; xf.lda = 16; xf.G = G; xf.D = D;
  RowSrc gf{}; gf.mode = SRC_DSEMF_OUT; gf.b = (const h16*)d_semo_in; gf.ldb = semo->in_pad;
  gf.w_row = w_row; gf.g = g_feat; gf.gw = D; row_src_rays(gf, N, S1, S2); gf.D = D; gf.fold_geo = fold_geo;
  return mlp_bwd_recomp_src(semf, xf, gf, rows, nullptr, d_semf_in, dW_semf, found_inf, s);
}
