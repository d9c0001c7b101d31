// Wide semantic heads (LSeg configuration: hidden_dim_semantic = 512, autolabel/models.py:117-136, docs/vision-language.md:19,
// scripts/ros/node.py:166-176): semantic_features 16 -> 512 -> 512 -> 512 and semantic_out 528 -> 64 -> C as hand-written
// MFMA GEMMs.  512 x 512 fp16 weights (512 KB per layer) do not fit the 160 KB LDS, so these heads cannot use the
// register-chained kernels of mlp.hip (weights resident in LDS, activations in registers); here every layer is one launch
// that streams the sample rows once, with the weight tiles re-read from L2 (1.1 MB for the whole head) and everything that
// used to be a separate pass fused into the prologue / epilogue:
//   * prologue: the layer input is built on the fly -- [geo_feat, 1] from the density head's output rows, relu(f) ++ [geo_feat, 1]
//     for semantic_out (models.py:254: cat[relu(features), geo_feat]) -- no [rows, 528] operand is ever materialised;
//   * epilogue: ReLU, ReLU' mask of the backward pass ((G W) * (act > 0)), accumulation into an existing gradient
//     (d f = d f_composite + mask * (dH W)), fp16 overflow watch.
// Two kernels:
//   k_wide_nt : Y[M, N]  = epi( A[M, K] W[N, K]^T )          forward layers and the data gradients (with W^T copies)
//   k_wide_tn : dW[N, K] += G[M, N]^T A[M, K]                weight gradients (contraction over the samples)
// v_mfma_f32_32x32x16_f16 throughout; operand tiles staged in LDS; k_wide_tn reads them with ds_read_b64_tr_b16.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) h16 lds_h16w;
typedef short s16x4w __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4w* lds_s16x4w_ptr;

__device__ inline f32x16 wmfma(h16x8 a, h16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// A-operand source: columns [0, K1) from a1 (optionally through ReLU), columns [K1, K1 + 16) = [geo_feat(1..15), 1] taken from
// the density head's output rows (sigma_out: [h0, geo1 .. geo15]) when geo != NULL.
struct WideSrc {
  const h16* a1; int lda1, K1, relu1;
  const h16* geo;   // sigma_out [M, 16] or NULL
  int G;            // geo_feat_dim (columns G .. 15 of the geo block are 1-padding)
};

__device__ inline h16x8 wide_chunk(const WideSrc& s, size_t row, int k0, int K) {
  h16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (h16)0.f;
  if (k0 >= K) return v;
  if (k0 < s.K1) {
    v = *(const h16x8*)(s.a1 + row * (size_t)s.lda1 + k0);
    if (s.relu1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > (h16)0.f ? v[j] : (h16)0.f;
    }
    return v;
  }
  const int j0 = k0 - s.K1;   // 0 or 8 within the geo block
  // columns j0 .. j0 + 7 of [geo1 .. geoG, 1, 1, ...] = halves 1 + j0 .. 8 + j0 of the 32-byte row: ONE 16-byte load and one more half
  // (round 4 read the eight halves one by one -- unaligned by one -- and a wave's 2-byte loads at a 32-byte stride cost 16-32 cache
  //  lines per instruction: the weight-gradient kernels, which rebuild the block for every 64-row tile, spent more L1 lookups on these
  //  16 columns than on their whole gradient tile)
  const h16* g = s.geo + row * 16;
  const h16x8 a = *(const h16x8*)(g + j0);
  const h16 nxt = j0 == 0 ? g[8] : (h16)1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = j0 + j;
    const h16 src = j < 7 ? a[j + 1] : nxt;
    v[j] = c < s.G ? src : (h16)1.0f;
  }
  return v;
}

struct WideNT {
  WideSrc a; int M, N, K;
  const h16* w; int ldw;        // [N, ldw] row-major, K columns used
  h16* y; int ldy;              // [M, ldy]
  int relu;                     // epilogue ReLU
  const h16* mask; int ldm;     // multiply by (mask[m][n] > 0) (ReLU' of the layer this gradient flows into) or NULL
  const h16* add; int lda;      // + add[m][n] before the store or NULL
  int* found_inf;               // raised when an output is not finite in fp16
  // "generated" first hidden layer (h1 = relu([geo_feat, 1] W0^T) is never stored: 1 GB at 2^20 rows x 512): the ReLU' mask of the
  // layer this gradient flows into is recomputed in the accumulator layout -- ONE matrix instruction per 32 x 32 block (K = 16) --
  // from the density head's output rows and W0 [N, 16] instead of being read back
  const h16* mgeo; const h16* mw0; int mG;
  // per 32-row tile sums of w[row] * y[row][n] (fp32 [ceil(M / 32)][N]): the compositing of a per-sample activation without a second
  // pass over its rows (k_wide_nt_gen<.., SUMS>; a tile must lie inside one ray: sample counts multiples of 32)
  const float* w_row; float* tsums;
};

// h1 = relu([geo_feat, 1] W0^T) on the fly.  wmfma(W0 fragment, geo fragment) leaves C[feature][sample]: lane = sample, register r
// = feature 8 (r / 4) + 4 hf + r % 4 of the 32-feature block -- registers 0..7 and 8..15 are, as they stand, the B operands of two
// 16-wide k-steps of the NEXT layer with the contraction index permuted inside every group of 16 (order 0-3, 8-11 | 4-7, 12-15):
// the weight matrix of that layer is stored with its columns in the same order (`wide_kperm`, pipeline.Params.wide_wp), so its
// fragments stay contiguous 16-byte reads.  A permutation of the contraction index changes nothing but the summation order.
typedef short ws16x2 __attribute__((ext_vector_type(2)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wu32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wu32x4 __attribute__((ext_vector_type(4)));
// fp16(relu(a)), fp16(relu(b)) in two instructions (mlp_shared.h relu2: v_cvt_pk_f16_f32, then max on the packed halves as signed integers)
__device__ inline uint32_t wrelu2(float a, float b) {
  union { h16x2 h; ws16x2 i; uint32_t w; } u; u.h = __builtin_convertvector((wf32x2){a, b}, h16x2);
  u.i = __builtin_elementwise_max(u.i, (ws16x2){0, 0});
  return u.w;
}
__device__ inline void wide_gen_pack(const f32x16& h, h16x8& lo, h16x8& hi) {
  const wu32x4 l = {wrelu2(h[0], h[1]), wrelu2(h[2], h[3]), wrelu2(h[4], h[5]), wrelu2(h[6], h[7])};
  const wu32x4 u = {wrelu2(h[8], h[9]), wrelu2(h[10], h[11]), wrelu2(h[12], h[13]), wrelu2(h[14], h[15])};
  lo = __builtin_bit_cast(h16x8, l); hi = __builtin_bit_cast(h16x8, u);
}
__device__ inline h16x8 wide_geo_chunk(const h16* geo, int G, size_t row, int hf) {
  const WideSrc s{nullptr, 0, 0, 0, geo, G};
  return wide_chunk(s, row, 8 * hf, 16);
}

// Epilogue shared by the two NT kernels.  In the accumulators a lane holds ONE sample row (register r of block b = output column
// 32 b + 8 (r / 4) + 4 hf + r % 4): storing from there writes 8-byte pieces to 32 different rows per instruction (measured: the
// 1M x 512 output cost more than the GEMM).  Each wave therefore turns its 32 x CG fp32 block through LDS and then works
// row-contiguously: 8 lanes cover 64 columns of a row, so mask / addend loads and the fp16 stores are 16 bytes per lane.
template <int BN, int MT, bool SUMS = false>
__device__ inline bool wide_epilogue(const WideNT& p, f32x16 (&acc)[MT][BN / 32], unsigned char* smem_w, int m0, int n0, int wave, int lane,
                              const float* wrow_lds = nullptr /* SUMS: the wave's MT x 32 row weights, staged in LDS by the kernel's prologue */) {
  constexpr int CG = BN < 64 ? BN : 64, EP = CG + 4;
  const int hf = lane >> 5, c = lane & 31;
  bool bad = false;
  __syncthreads();                       // every wave is done with the operand tiles: the space becomes the transpose buffer
  float* ep = (float*)smem_w + wave * (32 * EP);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int cg = 0; cg < BN / CG; ++cg) {
#pragma unroll
      for (int bb = 0; bb < CG / 32; ++bb) {
        const int b = cg * (CG / 32) + bb;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *(f32x4*)(ep + c * EP + bb * 32 + rg * 8 + hf * 4) = (f32x4){acc[t][b][rg * 4], acc[t][b][rg * 4 + 1], acc[t][b][rg * 4 + 2], acc[t][b][rg * 4 + 3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      constexpr int LPR = CG / 8;          // lanes per row (8 columns each)
      float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // SUMS: this lane's share of sum_rows w[row] * y[row][col .. col + 7]
#pragma unroll
      for (int it = 0; it < 32 / (64 / LPR); ++it) {
        const int r = it * (64 / LPR) + lane / LPR, col = (lane % LPR) * 8;
        const int m = m0 + (wave * MT + t) * 32 + r, n = n0 + cg * CG + col;
        if (m < p.M && n < p.N) {
          float v[8];
          const f32x4 v0 = *(const f32x4*)(ep + r * EP + col), v1 = *(const f32x4*)(ep + r * EP + col + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
          const bool full = n + 8 <= p.N;     // N is a multiple of 4: a row ends on a full or a half chunk
          if (p.mask) {
            const h16* mp = p.mask + (size_t)m * p.ldm + n;
            h16x4 m0v = *(const h16x4*)mp, m1v = full ? *(const h16x4*)(mp + 4) : (h16x4){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = m0v[j] > (h16)0.f ? v[j] : 0.f; v[4 + j] = m1v[j] > (h16)0.f ? v[4 + j] : 0.f; }
          }
          if (p.add) {
            const h16* ap = p.add + (size_t)m * p.lda + n;
            h16x4 a0 = *(const h16x4*)ap, a1 = full ? *(const h16x4*)(ap + 4) : (h16x4){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += (float)a0[j]; v[4 + j] += (float)a1[j]; }
          }
          h16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (p.relu) v[j] = fmaxf(v[j], 0.f);
            if (j < 4 || full) bad |= !(fabsf(v[j]) <= 65504.f);
            o[j] = (h16)v[j];
          }
          if constexpr (SUMS) {     // (the fp16 values as stored: what a pass over the rows would read)
            const float w = wrow_lds[t * 32 + r];     // (a global load here sat on the epilogue's critical path eight times per wave: +84 us)
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < 4 || full) ps[j] += w * (float)o[j];
          }
          h16* yp = p.y + (size_t)m * p.ldy + n;
          if (full && (p.ldy & 7) == 0) *(h16x8*)yp = o;               // one 16-byte store per lane
          else { *(h16x4*)yp = (h16x4){o[0], o[1], o[2], o[3]}; if (full) *(h16x4*)(yp + 4) = (h16x4){o[4], o[5], o[6], o[7]}; }
        }
      }
      if constexpr (SUMS) {   // fold the row groups (lanes LPR apart hold the same columns): a fixed tree, then one store per column group
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
          for (int j = 0; j < 8; ++j) ps[j] += __shfl_xor(ps[j], off);
        const int mrow = m0 + (wave * MT + t) * 32, n = n0 + cg * CG + (lane % LPR) * 8;
        if (lane < LPR && mrow < p.M && n < p.N) {
          float* dst = p.tsums + (size_t)(mrow / 32) * p.N + n;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (n + j < p.N) dst[j] = ps[j];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  return bad;
}

#define WNT_BK 64
#define WNT_PITCH (WNT_BK + 8)   // 144-byte rows: the 16-byte fragment reads of 16 consecutive rows fall into distinct 16-byte bank groups

// Block = 4 waves; wave w owns WM = 32 * MT sample rows x BN output columns (MT x BN/32 accumulator blocks): with MT = 2, BN = 128
// a k-step costs 6 fragment reads for 8 MFMAs (a 32 x 128 wave tile needs 5 for 4 and is LDS-bound).
template <int BN, int MT>
__global__ __launch_bounds__(256) void k_wide_nt(WideNT p) {
  constexpr int BM = 128 * MT, NB = BN / 32, XC = BM * (WNT_BK / 8) / 256, WC = (BN * (WNT_BK / 8) + 255) / 256;
  constexpr int CG = BN < 64 ? BN : 64, EP = CG + 4;      // epilogue: column groups of CG fp32 values per row, pitch EP floats
  constexpr int OPER = (BM + BN) * WNT_PITCH * 2, EPIL = 4 * 32 * EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_w[OPER > EPIL ? OPER : EPIL];
  h16* Xs = (h16*)smem_w;
  h16* Ws = Xs + BM * WNT_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, c = lane & 31;
  // XCD-aware tile order (block b runs on XCD b % 8, each XCD has its own L2): the N / BN column tiles of one row tile get ids
  // that differ by multiples of 8, so they land on the SAME XCD back to back and the row tile of A is fetched from HBM once
  // (with a plain 2-D grid every column tile re-read its 128 x K slice of A: 4 GB instead of 1 GB for the 512-wide layers)
  const int ntn = (p.N + BN - 1) / BN, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt_i = (jj / ntn) * 8 + xcd;
  if (mt_i * BM >= p.M) return;
  const int m0 = mt_i * BM, n0 = (jj % ntn) * BN;
  f32x16 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.f;
  const int nkt = (p.K + WNT_BK - 1) / WNT_BK;
  // global -> register staging of one k-tile (8 chunks of 8 halves per row)
  h16x8 xr[XC], wr[WC];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < XC; ++i) {
      const int ch = tid + 256 * i, r = ch >> 3, k0 = kt * WNT_BK + (ch & 7) * 8;
      const int m = m0 + r;
      if (m < p.M) xr[i] = wide_chunk(p.a, (size_t)m, k0, p.K);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[i][j] = (h16)0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < WC; ++i) {
      const int ch = tid + 256 * i, r = ch >> 3, k0 = kt * WNT_BK + (ch & 7) * 8;
      h16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (h16)0.f;
      if (ch < BN * 8 && n0 + r < p.N && k0 < p.K) v = *(const h16x8*)(p.w + (size_t)(n0 + r) * p.ldw + k0);
      wr[i] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < XC; ++i) { const int ch = tid + 256 * i; *(h16x8*)(Xs + (ch >> 3) * WNT_PITCH + (ch & 7) * 8) = xr[i]; }
#pragma unroll
    for (int i = 0; i < WC; ++i) {
      const int ch = tid + 256 * i;
      if (ch < BN * 8) *(h16x8*)(Ws + (ch >> 3) * WNT_PITCH + (ch & 7) * 8) = wr[i];
    }
  };
  fetch(0);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();            // the previous tile has been consumed
    stash();
    __syncthreads();
    if (kt + 1 < nkt) fetch(kt + 1);   // in flight while this tile is multiplied
#pragma unroll
    for (int ks = 0; ks < WNT_BK / 16; ++ks) {
      // B operand = sample rows, A operand = weight rows (output columns): the lane owns sample row c of its 32-row group
      h16x8 xb[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) xb[t] = *(const h16x8*)(Xs + ((wave * MT + t) * 32 + c) * WNT_PITCH + ks * 16 + hf * 8);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const h16x8 wa = *(const h16x8*)(Ws + (b * 32 + c) * WNT_PITCH + ks * 16 + hf * 8);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][b] = wmfma(wa, xb[t], acc[t][b]);
      }
    }
  }
  const bool bad = wide_epilogue<BN, MT>(p, acc, smem_w, m0, n0, wave, lane);
  if (bad && p.found_inf) *p.found_inf = 1;
}

// ---- plain-source variant with direct global -> LDS loads (global_load_lds_dwordx4: no staging registers, no ds_write pass).
// The DMA writes lane i of an instruction at (wave-uniform base + 16 i), so a tile is stored LINEAR, [rows][64 halves], and the
// bank-conflict swizzle sits on both sides instead: logical 16-byte chunk c of row r lives in slot c ^ (r & 7) -- the lane that
// fills slot s of row r fetches chunk s ^ (r & 7) from global memory, the fragment read of chunk c goes to slot c ^ (r & 7).
// Rows beyond M / N are clamped to the last valid row (their results are never stored).  Needs K % 64 == 0 and a plain A operand.
template <int BN, bool MASKGEN = false>
__global__ __launch_bounds__(256) void k_wide_nt_dma(WideNT p) {
  constexpr int BM = 128, NB = BN / 32, BK = 64, CG = BN < 64 ? BN : 64, EP = CG + 4;
  constexpr int OPER = (BM + BN) * BK * 2, EPIL = 4 * 32 * EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_w[OPER > EPIL ? OPER : EPIL];
  h16* Xs = (h16*)smem_w;
  h16* Ws = Xs + BM * BK;
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt_i = (jj / ntn) * 8 + xcd;
  if (mt_i * BM >= p.M) return;
  const int m0 = mt_i * BM, n0 = (jj % ntn) * BN;
  f32x16 acc[1][NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
  // per-lane source rows of this wave's DMA instructions (8 rows x 8 slots per instruction)
  const int sub = lane >> 3, slot = lane & 7;
  const h16* xsrc[4]; const h16* wsrc[NB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + sub;
    xsrc[i] = p.a.a1 + (size_t)min(m0 + row, p.M - 1) * p.a.lda1 + ((slot ^ (row & 7)) * 8);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + sub;
    wsrc[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.ldw + ((slot ^ (row & 7)) * 8);
  }
  const int nkt = p.K / BK;
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Xs + (wave * 4 + i) * 512), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Ws + (wave * NB + i) * 512), 16, 0, 0);
    __syncthreads();            // (the compiler drains vmcnt before the barrier: the tile has landed)
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int xr = wave * 32 + c;
      const h16x8 xb = *(const h16x8*)(Xs + xr * BK + (((ks * 2 + hf) ^ (xr & 7)) * 8));
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int wr_ = b * 32 + c;
        const h16x8 wa = *(const h16x8*)(Ws + wr_ * BK + (((ks * 2 + hf) ^ (wr_ & 7)) * 8));
        acc[0][b] = wmfma(wa, xb, acc[0][b]);
      }
    }
    __syncthreads();            // the tile has been consumed
  }
  if constexpr (MASKGEN) {   // ReLU' of the generated layer: acc *= (h1 > 0), h1 recomputed block by block in the accumulators' own layout
    const h16x8 gb = wide_geo_chunk(p.mgeo, p.mG, (size_t)min(m0 + wave * 32 + c, p.M - 1), hf);
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const h16x8 fa = *(const h16x8*)(p.mw0 + (size_t)min(n0 + b * 32 + c, p.N - 1) * 16 + 8 * hf);
      const f32x16 h = wmfma(fa, gb, zero);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {   // (h1 > 0 as stored: fp16 of the ReLU'd value)
        const uint32_t hp = wrelu2(h[r], h[r + 1]);
        acc[0][b][r] = (hp & 0xFFFFu) ? acc[0][b][r] : 0.f;
        acc[0][b][r + 1] = (hp >> 16) ? acc[0][b][r + 1] : 0.f;
      }
    }
  }
  const bool bad = wide_epilogue<BN, 1>(p, acc, smem_w, m0, n0, wave, lane);
  if (bad && p.found_inf) *p.found_inf = 1;
}

// Y = epi(relu([geo_feat, 1] W0^T) W^T): the first TWO layers of semantic_features in one launch (models.py:117-125).  The K = 16
// layer is one matrix instruction per 32 features and wave (12 % more matrix work than the 512 x 512 layer alone, redone by each of
// the N / 128 column tiles of a row tile); its 1 GB output, the launch that wrote it (414 us: the slowest of the forward pass) and
// this layer's 1 GB read are gone, and no sample tile passes through LDS at all -- the generated registers ARE the B operands.
// Weight tiles as in k_wide_nt_dma (global_load_lds, swizzled slots); `w` has its columns in wide_kperm order.  K % 64 == 0.
struct WideGen { const h16* geo; int G; const h16* w0; };
// MT row blocks of 32 samples per wave (block = 128 MT rows): with MT = 2 a k-step reads four weight fragments for eight matrix
// instructions -- half the LDS bytes per instruction of the 32-row wave tile, which sits at the LDS bandwidth (4 x 1 KB per 128 clk and
// SIMD) -- and there is no sample tile whose staging registers would push the 64-row tile past 256 VGPRs (k_wide_nt).
template <int BN, int MT, bool SUMS>
__global__ __launch_bounds__(256) void k_wide_nt_gen(WideNT p, WideGen g) {
  constexpr int BM = 128 * MT, NB = BN / 32, BK = 64, CG = BN < 64 ? BN : 64, EP = CG + 4;
  constexpr int OPER = 2 * BN * BK * 2, EPIL = 4 * 32 * EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_w[OPER > EPIL ? OPER : EPIL];
  h16* Ws = (h16*)smem_w;
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (p.N + BN - 1) / BN, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mt_i = (jj / ntn) * 8 + xcd;
  if (mt_i * BM >= p.M) return;
  const int m0 = mt_i * BM, n0 = (jj % ntn) * BN;
  f32x16 acc[MT][NB], zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[t][b] = zero;
  const int sub = lane >> 3, slot = lane & 7;
  const h16* wsrc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + sub;
    wsrc[i] = p.w + (size_t)min(n0 + row, p.N - 1) * p.ldw + ((slot ^ (row & 7)) * 8);
  }
  h16x8 gb[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) gb[t] = wide_geo_chunk(g.geo, g.G, (size_t)min(m0 + (wave * MT + t) * 32 + c, p.M - 1), hf);
  __shared__ float wrow_s[SUMS ? 4 * MT * 32 : 1];
  if constexpr (SUMS) {   // the wave's row weights: read by its own epilogue only (the barriers of the main loop lie in between)
    if (hf == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t) { const int m = m0 + (wave * MT + t) * 32 + c; wrow_s[(wave * MT + t) * 32 + c] = m < p.M ? p.w_row[m] : 0.f; }
    }
  }
  const int nkt = p.K / BK;
  // the generated operands run one 32-feature block ahead of the matrix instructions that consume them
  h16x8 xb[2][MT][2];
  auto gen = [&](int fb, int buf) {
    const h16x8 fa = *(const h16x8*)(g.w0 + (size_t)(fb * 32 + c) * 16 + 8 * hf);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const f32x16 h = wmfma(fa, gb[t], zero);
      wide_gen_pack(h, xb[buf][t][0], xb[buf][t][1]);
    }
  };
  gen(0, 0);
  // weight tiles double-buffered: the tile of k-step kt + 1 is in flight while kt is multiplied (ONE barrier per tile; with a single
  // buffer the whole load latency sat between two barriers and only other blocks' waves could hide it).  Buffer indices are
  // compile-time constants (two k-tiles per loop trip).
  auto load_w = [&](int kt, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(Ws + buf * (BN * BK) + (wave * NB + i) * 512), 16, 0, 0);
  };
  auto tile = [&](int kt, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    __syncthreads();            // (vmcnt drained before the barrier: tile kt has landed; every wave is done with the other buffer)
    if (kt + 1 < nkt) load_w(kt + 1, std::integral_constant<int, buf ^ 1>{});
    const h16* Wt = Ws + buf * (BN * BK);
#pragma unroll
    for (int f2 = 0; f2 < 2; ++f2) {
      const int fb = 2 * kt + f2;
      if (fb + 1 < 2 * nkt) gen(fb + 1, (f2 + 1) & 1);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * f2 + k2;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int wr_ = b * 32 + c;
          const h16x8 wa = *(const h16x8*)(Wt + wr_ * BK + (((ks * 2 + hf) ^ (wr_ & 7)) * 8));
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][b] = wmfma(wa, xb[f2 & 1][t][k2], acc[t][b]);
        }
      }
    }
  };
  load_w(0, std::integral_constant<int, 0>{});
  for (int kt = 0; kt < nkt; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nkt) tile(kt + 1, std::integral_constant<int, 1>{});
  }
  const bool bad = wide_epilogue<BN, MT, SUMS>(p, acc, smem_w, m0, n0, wave, lane, &wrow_s[wave * MT * 32]);
  if (bad && p.found_inf) *p.found_inf = 1;
}

extern "C" int aln_wide_nt(const void* a1, int32_t lda1, int32_t K1, int32_t relu1, const void* geo, int32_t G, int32_t M, int32_t N,
                           const void* w, int32_t ldw, void* y, int32_t ldy, int32_t relu, const void* mask, int32_t ldm,
                           const void* add, int32_t lda, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(w && y && (a1 || geo) && M >= 0 && N > 0, "wide_nt: bad arguments");
  ALN_REQUIRE(K1 % 8 == 0 && N % 4 == 0 && ldw % 8 == 0 && (!a1 || lda1 % 8 == 0) && ldy % 4 == 0, "wide_nt: K1 / N / leading dimensions must be multiples of 8 / 4");
  ALN_REQUIRE((!mask || ldm % 4 == 0) && (!add || lda % 4 == 0), "wide_nt: mask / addend leading dimensions must be multiples of 4");
  if (M == 0) return 0;
  WideNT p;
  p.a = WideSrc{(const h16*)a1, lda1, a1 ? K1 : 0, relu1, (const h16*)geo, G};
  p.M = M; p.N = N; p.K = p.a.K1 + (geo ? 16 : 0);
  p.w = (const h16*)w; p.ldw = ldw; p.y = (h16*)y; p.ldy = ldy; p.relu = relu;
  p.mask = (const h16*)mask; p.ldm = ldm; p.add = (const h16*)add; p.lda = lda; p.found_inf = found_inf;
  p.mgeo = nullptr; p.mw0 = nullptr; p.mG = 0; p.w_row = nullptr; p.tsums = nullptr;
  ALN_REQUIRE(ldw >= p.K, "wide_nt: weight rows shorter than K");
  hipStream_t s = (hipStream_t)stream;
  // (measured on the 1M x 512 x 512 layers: wave tile 32 x 128 at 2 waves/SIMD 1.3 ms; 64 x 128 needs > 256 VGPRs, one wave per
  //  SIMD: 1.5 ms)
  auto grid = [&](int bm, int bn) { return dim3((unsigned)(((M + bm - 1) / bm + 7) / 8 * 8 * ((N + bn - 1) / bn))); };
  if (a1 && !geo && !relu1 && p.K % 64 == 0 && ((uintptr_t)a1 & 15) == 0 && ((uintptr_t)w & 15) == 0) {
    // plain operands: tiles filled by global_load_lds (k_wide_nt_dma)
    if (N > 64) hipLaunchKernelGGL((k_wide_nt_dma<128>), grid(128, 128), dim3(256), 0, s, p);
    else if (N > 32) hipLaunchKernelGGL((k_wide_nt_dma<64>), grid(128, 64), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_wide_nt_dma<32>), grid(128, 32), dim3(256), 0, s, p);
    ALN_CHECK_LAUNCH("wide_nt_dma");
    return 0;
  }
  if (N > 64) hipLaunchKernelGGL((k_wide_nt<128, 1>), grid(128, 128), dim3(256), 0, s, p);
  else if (N > 32) hipLaunchKernelGGL((k_wide_nt<64, 1>), grid(128, 64), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((k_wide_nt<32, 1>), grid(128, 32), dim3(256), 0, s, p);
  ALN_CHECK_LAUNCH("wide_nt");
  return 0;
}

// Y[M, N] = epi(relu([geo_feat, 1] W0[K, 16]^T) W[N, K]^T): semantic_features' first two layers in one launch, the first one generated
// (k_wide_nt_gen).  `w_perm`: W with the columns of every group of 16 in wide_kperm order (0-3, 8-11, 4-7, 12-15).
// `tile_sums` (optional, with `w_row`): fp32 [ceil(M / 32)][N], tile t = sum over rows 32 t .. 32 t + 31 of w_row[row] * Y[row][:].
extern "C" int aln_wide_nt_gen(const void* geo, int32_t G, const void* w0, int32_t M, int32_t N, int32_t K, const void* w_perm, int32_t ldw,
                               void* y, int32_t ldy, int32_t relu, const float* w_row, float* tile_sums, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(!tile_sums || w_row, "wide_nt_gen: tile sums need the row weights");
  ALN_REQUIRE(geo && w0 && w_perm && y && M >= 0 && N > 0, "wide_nt_gen: bad arguments");
  ALN_REQUIRE(K > 0 && K % 64 == 0 && N % 4 == 0 && ldw % 8 == 0 && ldw >= K && ldy % 4 == 0, "wide_nt_gen: K must be a multiple of 64, N / leading dimensions of 4 / 8");
  ALN_REQUIRE(((uintptr_t)w_perm & 15) == 0 && ((uintptr_t)w0 & 15) == 0 && ((uintptr_t)geo & 15) == 0, "wide_nt_gen: operands must be 16-byte aligned");
  if (M == 0) return 0;
  WideNT p;
  p.a = WideSrc{nullptr, 0, 0, 0, nullptr, 0};
  p.M = M; p.N = N; p.K = K; p.w = (const h16*)w_perm; p.ldw = ldw; p.y = (h16*)y; p.ldy = ldy; p.relu = relu;
  p.mask = nullptr; p.ldm = 0; p.add = nullptr; p.lda = 0; p.found_inf = found_inf; p.mgeo = nullptr; p.mw0 = nullptr; p.mG = 0;
  p.w_row = w_row; p.tsums = tile_sums;
  const WideGen g{(const h16*)geo, G, (const h16*)w0};
  auto grid = [&](int bm, int bn) { return dim3((unsigned)(((M + bm - 1) / bm + 7) / 8 * 8 * ((N + bn - 1) / bn))); };
  // (MT = 2, a 64-row wave tile at two waves per SIMD, measured slower in the step: 711 vs 669 us at 2^20 rows)
  if (tile_sums) {
    ALN_REQUIRE(N > 64, "wide_nt_gen: tile sums are instantiated for the 128-column tile only");
    hipLaunchKernelGGL((k_wide_nt_gen<128, 1, true>), grid(128, 128), dim3(256), 0, (hipStream_t)stream, p, g);
  } else if (N > 64) hipLaunchKernelGGL((k_wide_nt_gen<128, 1, false>), grid(128, 128), dim3(256), 0, (hipStream_t)stream, p, g);
  else if (N > 32) hipLaunchKernelGGL((k_wide_nt_gen<64, 1, false>), grid(128, 64), dim3(256), 0, (hipStream_t)stream, p, g);
  else hipLaunchKernelGGL((k_wide_nt_gen<32, 1, false>), grid(128, 32), dim3(256), 0, (hipStream_t)stream, p, g);
  ALN_CHECK_LAUNCH("wide_nt_gen");
  return 0;
}
// Y[M, N] = (A[M, K] W[N, K]^T) * (h1 > 0) with h1 = relu([geo_feat, 1] W0[N, 16]^T) recomputed in the accumulators (no mask rows are
// read): the data gradient that flows into the generated layer.  Plain A operand, K % 64 == 0.
extern "C" int aln_wide_nt_maskgen(const void* a1, int32_t lda1, int32_t M, int32_t N, int32_t K, const void* w, int32_t ldw, void* y,
                                   int32_t ldy, const void* geo, int32_t G, const void* w0, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(a1 && w && y && geo && w0 && M >= 0 && N > 0, "wide_nt_maskgen: bad arguments");
  ALN_REQUIRE(K > 0 && K % 64 == 0 && N % 4 == 0 && ldw % 8 == 0 && ldw >= K && lda1 % 8 == 0 && ldy % 4 == 0, "wide_nt_maskgen: K must be a multiple of 64, N / leading dimensions of 4 / 8");
  ALN_REQUIRE(((uintptr_t)a1 & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)w0 & 15) == 0, "wide_nt_maskgen: operands must be 16-byte aligned");
  if (M == 0) return 0;
  WideNT p;
  p.a = WideSrc{(const h16*)a1, lda1, K, 0, nullptr, 0};
  p.M = M; p.N = N; p.K = K; p.w = (const h16*)w; p.ldw = ldw; p.y = (h16*)y; p.ldy = ldy; p.relu = 0;
  p.mask = nullptr; p.ldm = 0; p.add = nullptr; p.lda = 0; p.found_inf = found_inf;
  p.mgeo = (const h16*)geo; p.mw0 = (const h16*)w0; p.mG = G; p.w_row = nullptr; p.tsums = nullptr;
  auto grid = [&](int bm, int bn) { return dim3((unsigned)(((M + bm - 1) / bm + 7) / 8 * 8 * ((N + bn - 1) / bn))); };
  if (N > 64) hipLaunchKernelGGL((k_wide_nt_dma<128, true>), grid(128, 128), dim3(256), 0, (hipStream_t)stream, p);
  else if (N > 32) hipLaunchKernelGGL((k_wide_nt_dma<64, true>), grid(128, 64), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((k_wide_nt_dma<32, true>), grid(128, 32), dim3(256), 0, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("wide_nt_maskgen");
  return 0;
}

// ---------------------------------------------------------------- weight gradients
struct WideTN {
  WideSrc a; int M, N, K;
  const h16* g; int ldg;     // [M, ldg], N columns used
  float* ws;                 // [slabs][N][K] fp32 partial sums, one slab per block row range; k_wide_dw_reduce adds them to dW in a fixed order
  int slab;                  // sample rows per block
  int tn, tk;                // output tiles in n and k
};

#define WTN_BM 64            // sample rows per LDS tile
#define WTN_TN 128           // tile edge in n  (2 x 2 waves of 64 n x 128 k)
#define WTN_TK 256           // tile edge in k
#define WTN_PG (WTN_TN + 24) // halves (304 B: rows stay 16-byte aligned for the staging stores)
#define WTN_PA (WTN_TK + 24)

template <int PITCH>
struct WTile {
  lds_h16w* p;
  __device__ inline lds_h16w* at(int row, int col) const { return p + row * PITCH + col; }
};
template <class T>
__device__ inline h16x8 wtr_frag(T t, int col0, int ks, int lane) {
  // operand fragment for mfma 32x32x16: lane (i = lane & 31, hf = lane >> 5) gets tile[16 ks + 8 hf + 0..7][col0 + i]
  const int hf = lane >> 5;
  const int row = 16 * ks + 8 * hf + ((lane & 15) >> 2);
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w_ptr)t.at(row, col));
  s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w_ptr)t.at(row + 4, col));
  union { struct { s16x4w l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}

// Wave layout NWN x NWK (= 4 waves), every wave a 64 (n) x 128 (k) tile: 2 x 2 for the square layers (tile 128 x 256); 4 x 1 (tile
// 256 x 128) for an operand of at most 128 columns -- the 16-wide [geo_feat, 1] block of the first layer, whose 2 x 2 launch left
// the two waves of the second k half without a single column (745 us for a 17 GFLOP product, round 4).
template <int NWN, int NWK>
__global__ __launch_bounds__(256) void k_wide_tn(WideTN p) {
  constexpr int TN = 64 * NWN, TK = 128 * NWK, PG = TN + 24, PA = TK + 24, GC = TN / 32, AC = TK / 32;
  __shared__ __attribute__((aligned(16))) h16 Gs[WTN_BM * PG];
  __shared__ __attribute__((aligned(16))) h16 As[WTN_BM * PA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, c = lane & 31;
  // XCD-aware order (see k_wide_nt): the tn x tk output tiles of one sample slab share an XCD, so the slab's rows of G and A
  // come from HBM once
  const int ntile = p.tn * p.tk, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int slab_i = (jj / ntile) * 8 + xcd, tile_i = jj % ntile;
  if (slab_i * p.slab >= p.M) return;
  const int n0 = (tile_i % p.tn) * TN, k0 = (tile_i / p.tn) * TK;
  const int mlo = slab_i * p.slab, mhi = min(p.M, mlo + p.slab);
  const int wn = (wave / NWK) * 64, wk = (wave % NWK) * 128;   // wave tile 64 (n) x 128 (k): 6 fragment reads per 8 MFMAs
  // 32-column blocks of this wave's k range that exist at all (the 16-wide geo_feat operand of the first layer fills ONE block)
  const int nbj = __builtin_amdgcn_readfirstlane(max(0, min(4, (p.K - k0 - wk + 31) / 32)));
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const WTile<PG> tG{(lds_h16w*)Gs};
  const WTile<PA> tA{(lds_h16w*)As};
  // G tile: 64 rows x TN / 8 chunks (GC per thread); A tile: 64 rows x TK / 8 chunks (AC per thread)
  h16x8 gr[GC], ar[AC];
  auto fetch = [&](int mt) {
    h16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (h16)0.f;
#pragma unroll
    for (int i = 0; i < GC; ++i) {
      const int ch = tid + 256 * i, r = ch / (TN / 8), cc = (ch % (TN / 8)) * 8;
      gr[i] = z;
      if (mt + r < mhi && n0 + cc < p.N) gr[i] = *(const h16x8*)(p.g + (size_t)(mt + r) * p.ldg + n0 + cc);
    }
#pragma unroll
    for (int i = 0; i < AC; ++i) {
      const int ch = tid + 256 * i, r = ch / (TK / 8), cc = (ch % (TK / 8)) * 8;
      ar[i] = z;
      if (mt + r < mhi) ar[i] = wide_chunk(p.a, (size_t)(mt + r), k0 + cc, p.K);
    }
  };
  fetch(mlo);
  for (int mt = mlo; mt < mhi; mt += WTN_BM) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < GC; ++i) { const int ch = tid + 256 * i; *(h16x8*)(Gs + (ch / (TN / 8)) * PG + (ch % (TN / 8)) * 8) = gr[i]; }
#pragma unroll
    for (int i = 0; i < AC; ++i) { const int ch = tid + 256 * i; *(h16x8*)(As + (ch / (TK / 8)) * PA + (ch % (TK / 8)) * 8) = ar[i]; }
    __syncthreads();
    if (mt + WTN_BM < mhi) fetch(mt + WTN_BM);
#pragma unroll
    for (int ks = 0; ks < WTN_BM / 16; ++ks) {
      h16x8 ga[2], ab[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) ga[i] = wtr_frag(tG, wn + 32 * i, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) if (j < nbj) ab[j] = wtr_frag(tA, wk + 32 * j, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nbj) acc[i][j] = wmfma(ga[i], ab[j], acc[i][j]);
    }
  }
  // lane holds column k = k0 + wk + 32 j + c; register r of block (i, j) is row n = n0 + wn + 32 i + 8 (r / 4) + 4 hf + r % 4
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk + 32 * j + c;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {   // the block's part of its slab of partial sums [N][K]: plain stores, every element written once
        const int n = n0 + wn + 32 * i + 8 * (r >> 2) + 4 * hf + (r & 3);
        if (n < p.N) p.ws[((size_t)slab_i * p.N + n) * p.K + k] = acc[i][j][r];
      }
    }
}

// dW[n][k] += sum over the slabs of partial sums, slab 0 first: fixed order, no atomics (the fp32 atomic flush of rounds 1-2 made the
// weight gradients of the LSeg-width heads -- and with them every LSeg training run -- depend on the order blocks finished in)
__global__ __launch_bounds__(256) void k_wide_dw_reduce(const float* __restrict__ ws, int slabs, int N, int K, float* __restrict__ dw, int lddw) {
  const size_t nk = (size_t)N * K;
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nk; e += (size_t)gridDim.x * 256) {
    float acc = 0.f;
    int sI = 0;
    for (; sI + 8 <= slabs; sI += 8) {   // eight loads in flight, added in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(sI + u) * nk + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sI < slabs; ++sI) acc += ws[(size_t)sI * nk + e];
    dw[(e / K) * (size_t)lddw + e % K] += acc;
  }
}
static bool wide_tn_narrow(int K) { return K <= 128; }   // 4 x 1 wave layout (tile 256 x 128) instead of 2 x 2 (128 x 256)
static void wide_tn_split(int M, int N, int K, int& tn, int& tk, int& slab, int& slabs) {
  const int TN = wide_tn_narrow(K) ? 256 : WTN_TN, TK = wide_tn_narrow(K) ? 128 : WTN_TK;
  tn = (N + TN - 1) / TN; tk = (K + TK - 1) / TK;
  // enough slabs to fill the chip (>= ~1024 blocks), each a multiple of the 64-row tile
  slabs = (1024 + tn * tk - 1) / (tn * tk);
  slab = ((M + slabs - 1) / slabs + WTN_BM - 1) / WTN_BM * WTN_BM;
  if (slab < WTN_BM) slab = WTN_BM;
  slabs = (M + slab - 1) / slab;
}
extern "C" int64_t aln_wide_tn_ws_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int tn, tk, slab, slabs;
  wide_tn_split(M, N, K, tn, tk, slab, slabs);
  return (int64_t)slabs * N * K * (int64_t)sizeof(float);
}
extern "C" int aln_wide_tn(const void* g, int32_t ldg, const void* a1, int32_t lda1, int32_t K1, int32_t relu1, const void* geo,
                           int32_t G, int32_t M, int32_t N, float* dw, int32_t lddw, void* ws, void* stream) {
  ALN_REQUIRE(g && dw && ws && (a1 || geo) && M >= 0 && N > 0, "wide_tn: bad arguments");
  ALN_REQUIRE(ldg % 8 == 0 && N % 8 == 0 && K1 % 8 == 0 && (!a1 || lda1 % 8 == 0), "wide_tn: N / K1 / leading dimensions must be multiples of 8");
  if (M == 0) return 0;
  WideTN p;
  p.a = WideSrc{(const h16*)a1, lda1, a1 ? K1 : 0, relu1, (const h16*)geo, G};
  p.M = M; p.N = N; p.K = p.a.K1 + (geo ? 16 : 0);
  p.g = (const h16*)g; p.ldg = ldg; p.ws = (float*)ws;
  ALN_REQUIRE(lddw >= p.K, "wide_tn: gradient rows shorter than K");
  int tn, tk, slab, slabs;
  wide_tn_split(M, N, p.K, tn, tk, slab, slabs);
  p.slab = slab; p.tn = tn; p.tk = tk;
  if (wide_tn_narrow(p.K)) hipLaunchKernelGGL((k_wide_tn<4, 1>), dim3((unsigned)((slabs + 7) / 8 * 8 * tn * tk)), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((k_wide_tn<2, 2>), dim3((unsigned)((slabs + 7) / 8 * 8 * tn * tk)), dim3(256), 0, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("wide_tn");
  const int64_t nk = (int64_t)N * p.K;
  hipLaunchKernelGGL(k_wide_dw_reduce, dim3((unsigned)((nk + 255) / 256 < 2048 ? (nk + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)ws, slabs, N, p.K, dw, lddw);
  ALN_CHECK_LAUNCH("wide_dw_reduce");
  return 0;
}

// dW[N, K] += G[M, N]^T relu([geo_feat, 1] W0[K, 16]^T): the weight gradient of the layer BEHIND the generated one.  The A operand (h1,
// 1 GB at 2^20 rows) is neither read nor staged: per 32 samples and 32 features ONE matrix instruction leaves h1^T in the
// accumulator layout (lane = feature, registers = samples 8 (r / 4) + 4 hf + r % 4), whose halves are the B operands of two 16-sample
// k-steps with the SAMPLE index permuted inside every group of 16; the G fragments take the same sample order through the row
// offsets of their transposing reads.  Same tiles, slabs and fixed-order reduction as k_wide_tn; a third of its LDS reads, 25 % more
// matrix instructions (both waves of a k half generate the same block).  Measured in the LSeg step (2^20 rows): 771 us, against 728
// for k_wide_tn reading a stored h1 and 818 for a form that generates the A tile into LDS once per block -- the stored form is
// the faster KERNEL; what pays is that h1 no longer exists (the launch that wrote it: 414 us, and its read by the second layer).
template <class T>
__device__ inline h16x8 wtr_frag_perm(T t, int col0, int ks, int lane) {
  const int hf = lane >> 5;
  const int row = 16 * ks + 4 * hf + ((lane & 15) >> 2);
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w_ptr)t.at(row, col));
  s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w_ptr)t.at(row + 8, col));
  union { struct { s16x4w l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}
__global__ __launch_bounds__(256) void k_wide_tn_gen(WideTN p, WideGen g) {
  __shared__ __attribute__((aligned(16))) h16 Gs[WTN_BM * WTN_PG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, c = lane & 31;
  const int ntile = p.tn * p.tk, xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int slab_i = (jj / ntile) * 8 + xcd, tile_i = jj % ntile;
  if (slab_i * p.slab >= p.M) return;
  const int n0 = (tile_i % p.tn) * WTN_TN, k0 = (tile_i / p.tn) * WTN_TK;
  const int mlo = slab_i * p.slab, mhi = min(p.M, mlo + p.slab);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
  f32x16 acc[2][4], zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero;
  const WTile<WTN_PG> tG{(lds_h16w*)Gs};
  h16x8 w0f[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w0f[j] = *(const h16x8*)(g.w0 + (size_t)min(k0 + wk + 32 * j + c, p.K - 1) * 16 + 8 * hf);
  h16x8 gr[4], ga_[2];
  auto fetch = [&](int mt) {
    h16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (h16)0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = tid + 256 * i, r = ch >> 4, cc = (ch & 15) * 8;
      gr[i] = z;
      if (mt + r < mhi && n0 + cc < p.N) gr[i] = *(const h16x8*)(p.g + (size_t)(mt + r) * p.ldg + n0 + cc);
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int m = mt + 32 * sb + c;
      ga_[sb] = m < mhi ? wide_geo_chunk(g.geo, g.G, (size_t)m, hf) : z;
    }
  };
  fetch(mlo);
  for (int mt = mlo; mt < mhi; mt += WTN_BM) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ch = tid + 256 * i; *(h16x8*)(Gs + (ch >> 4) * WTN_PG + (ch & 15) * 8) = gr[i]; }
    const h16x8 gcur[2] = {ga_[0], ga_[1]};
    __syncthreads();
    if (mt + WTN_BM < mhi) fetch(mt + WTN_BM);
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      h16x8 ab[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x16 h = wmfma(gcur[sb], w0f[j], zero);   // C[sample][feature]: lane = feature, registers = samples
        wide_gen_pack(h, ab[j][0], ab[j][1]);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * sb + k2;
        h16x8 ga[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[i] = wtr_frag_perm(tG, wn + 32 * i, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = wmfma(ga[i], ab[j][k2], acc[i][j]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk + 32 * j + c;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn + 32 * i + 8 * (r >> 2) + 4 * hf + (r & 3);
        if (n < p.N) p.ws[((size_t)slab_i * p.N + n) * p.K + k] = acc[i][j][r];
      }
    }
}
extern "C" int aln_wide_tn_gen(const void* g, int32_t ldg, const void* geo, int32_t G, const void* w0, int32_t M, int32_t N, int32_t K,
                               float* dw, int32_t lddw, void* ws, void* stream) {
  ALN_REQUIRE(g && dw && ws && geo && w0 && M >= 0 && N > 0, "wide_tn_gen: bad arguments");
  ALN_REQUIRE(ldg % 8 == 0 && N % 8 == 0 && K > 0 && K % 32 == 0 && lddw >= K, "wide_tn_gen: N / K / leading dimensions must be multiples of 8 / 32");
  ALN_REQUIRE(((uintptr_t)w0 & 15) == 0 && ((uintptr_t)g & 15) == 0, "wide_tn_gen: operands must be 16-byte aligned");
  // k_wide_tn_gen has the 2 x 2 wave layout (128 x 256 tiles) only; for K <= 128 wide_tn_split sizes the grid for 256 x 128 tiles, one
  // block per 256 output rows -- with N > 128 the columns [128, N) of the slabs would never be written
  ALN_REQUIRE(!wide_tn_narrow(K) || N <= 128, "wide_tn_gen: N = %d > 128 with K = %d <= 128 is not supported (the generated-operand kernel has the 128 x 256 tile only)", N, K);
  if (M == 0) return 0;
  WideTN p;
  p.a = WideSrc{nullptr, 0, 0, 0, nullptr, 0};
  p.M = M; p.N = N; p.K = K; p.g = (const h16*)g; p.ldg = ldg; p.ws = (float*)ws;
  int tn, tk, slab, slabs;
  wide_tn_split(M, N, K, tn, tk, slab, slabs);
  p.slab = slab; p.tn = tn; p.tk = tk;
  const WideGen gg{(const h16*)geo, G, (const h16*)w0};
  hipLaunchKernelGGL(k_wide_tn_gen, dim3((unsigned)((slabs + 7) / 8 * 8 * tn * tk)), dim3(256), 0, (hipStream_t)stream, p, gg);
  ALN_CHECK_LAUNCH("wide_tn_gen");
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(k_wide_dw_reduce, dim3((unsigned)((nk + 255) / 256 < 2048 ? (nk + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)ws, slabs, N, K, dw, lddw);
  ALN_CHECK_LAUNCH("wide_dw_reduce");
  return 0;
}

// The two consumers of the data gradient dh1 [M, N] that flows into the generated layer, in ONE pass over it:
//   d_fin[M, 16] = dh1 W0            (the gradient of the layer's 16 inputs: geo_feat + the constant one; A = W0^T [16, N])
//   dW0[N, 16]  += dh1^T [geo_feat, 1]
// As two launches (k_wide_nt_dma<32>, k_wide_tn<4, 1>) they read the 1 GB of dh1 twice (190 + 270 us in the LSeg step).  Here a block
// stages a 64-row x N tile once (N <= 512: 67 KB of LDS); every wave multiplies its 128 columns' transposed fragments with the
// [geo_feat, 1] rows for dW0 (k_wide_tn's scheme), and the tile's row-major fragments with W0^T for d_fin: wave w takes sample block
// w & 1 and column half w >> 1, the two halves of a sample block meet in LDS in a fixed order.  Slabs of rows, fixed-order reduction
// of the dW0 partial sums (k_wide_dw_reduce): bit-reproducible.
struct WideTNDin {
  const h16* g; int ldg; const h16* geo; int G; const h16* w0t; int ldw0t;   // dh1 [M, N], sigma_out [M, 16], W0^T [16, N]
  int M, N, slab; float* ws; h16* d_fin; int* found_inf;
};
#define WTD_PG (512 + 24)
__global__ __launch_bounds__(256) void k_wide_tn_din(WideTNDin p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
  h16* Gs = (h16*)smem_d;                        // [64][WTD_PG]
  h16* As = Gs + WTN_BM * WTD_PG;                // [64][24]   [geo_feat, 1] rows
  float* red = (float*)(As + WTN_BM * 24);       // [2 sample blocks][8 registers][64 lanes]: the upper column half's d_fin partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, c = lane & 31;
  const int slab_i = blockIdx.x;
  const int mlo = slab_i * p.slab, mhi = min(p.M, mlo + p.slab);
  if (mlo >= p.M) return;
  const int wn = wave * 128, sb = wave & 1, kh = wave >> 1, NH = p.N / 2;
  f32x16 acc[4], dacc, zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = zero;
  const WTile<WTD_PG> tG{(lds_h16w*)Gs};
  const WTile<24> tA{(lds_h16w*)As};
  const int nks = NH / 16;                        // 16-column steps of this wave's column half (16 at N = 512)
  // W0^T fragments of the column half (A operand: lane = input j, 16 real rows of 32; 8 columns per half-wave): 16 KB in all, re-read
  // from L1 / L2 for every tile (resident they cost 64 registers and the second block per CU)
  const h16* const wrow = p.w0t + (size_t)min(c, 15) * p.ldw0t + kh * NH + 8 * hf;
  h16x8 gr[16], ar;
  const WideSrc gsrc{nullptr, 0, 0, 0, p.geo, p.G};
  auto fetch = [&](int mt) {
    h16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (h16)0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ch = tid + 256 * i, r = ch >> 6, cc = (ch & 63) * 8;
      gr[i] = z;
      if (mt + r < mhi && cc < p.N) gr[i] = *(const h16x8*)(p.g + (size_t)(mt + r) * p.ldg + cc);
    }
    ar = z;
    if (tid < 128 && mt + (tid >> 1) < mhi) ar = wide_chunk(gsrc, (size_t)(mt + (tid >> 1)), 8 * (tid & 1), 16);
  };
  auto finish = [&](int mt) {   // d_fin rows of tile mt: lower column half (in registers) + upper half (in LDS), lower first
    if (kh == 0) {
      const int m = mt + 32 * sb + c;
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = dacc[r] + red[(sb * 8 + r) * 64 + lane];
      if (m < mhi) {   // register r = input j = 8 (r / 4) + 4 hf + r % 4
        bool bad = false;
        h16x4 o0, o1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] = (h16)v[r]; o1[r] = (h16)v[4 + r]; bad |= !(fabsf(v[r]) <= 65504.f) | !(fabsf(v[4 + r]) <= 65504.f); }
        *(h16x4*)(p.d_fin + (size_t)m * 16 + 4 * hf) = o0;
        *(h16x4*)(p.d_fin + (size_t)m * 16 + 8 + 4 * hf) = o1;
        if (bad && p.found_inf) *p.found_inf = 1;
      }
    }
  };
  fetch(mlo);
  bool pending = false;
  for (int mt = mlo; mt < mhi; mt += WTN_BM) {
    __syncthreads();                 // the previous tile is consumed; its upper-half partial sums are in `red`
    if (pending) finish(mt - WTN_BM);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int ch = tid + 256 * i; *(h16x8*)(Gs + (ch >> 6) * WTD_PG + (ch & 63) * 8) = gr[i]; }
    if (tid < 128) *(h16x8*)(As + (tid >> 1) * 24 + 8 * (tid & 1)) = ar;
    __syncthreads();
    if (mt + WTN_BM < mhi) fetch(mt + WTN_BM);
    // dW0: this wave's 128 columns x 16 inputs, contraction over the tile's 64 samples
#pragma unroll
    for (int ks = 0; ks < WTN_BM / 16; ++ks) {
      const h16x8 ab = wtr_frag(tA, 0, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = wmfma(wtr_frag(tG, wn + 32 * i, ks, lane), ab, acc[i]);
    }
    // d_fin: sample block sb, column half kh
    dacc = zero;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      if (kk < nks) dacc = wmfma(*(const h16x8*)(wrow + 16 * kk), *(const h16x8*)(Gs + (32 * sb + c) * WTD_PG + kh * NH + 16 * kk + 8 * hf), dacc);
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) red[(sb * 8 + r) * 64 + lane] = dacc[r];
    }
    pending = true;
  }
  __syncthreads();
  if (pending) finish(mlo + (mhi - mlo - 1) / WTN_BM * WTN_BM);
  // dW0 partial sums of the slab: lane = input k (16 real), register r of block i = column n = wn + 32 i + 8 (r / 4) + 4 hf + r % 4
  if (c < 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn + 32 * i + 8 * (r >> 2) + 4 * hf + (r & 3);
        if (n < p.N) p.ws[((size_t)slab_i * p.N + n) * 16 + c] = acc[i][r];
      }
  }
}
static void wide_tn_din_split(int M, int& slab, int& slabs) {
  slabs = 512;    // two blocks per CU; the partial sums the reduction reads grow with the slab count (1024 slabs: +26 us)
  slab = ((M + slabs - 1) / slabs + WTN_BM - 1) / WTN_BM * WTN_BM;
  if (slab < WTN_BM) slab = WTN_BM;
  slabs = (M + slab - 1) / slab;
}
extern "C" int64_t aln_wide_tn_din_ws_bytes(int32_t M, int32_t N) {
  if (M <= 0 || N <= 0) return 0;
  int slab, slabs;
  wide_tn_din_split(M, slab, slabs);
  return (int64_t)slabs * N * 16 * (int64_t)sizeof(float);
}
extern "C" int aln_wide_tn_din(const void* g, int32_t ldg, const void* geo, int32_t G, const void* w0t, int32_t ldw0t, int32_t M, int32_t N,
                               float* dw, int32_t lddw, void* ws, void* d_fin, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(g && geo && w0t && dw && ws && d_fin && M >= 0, "wide_tn_din: bad arguments");
  ALN_REQUIRE(N > 0 && N <= 512 && N % 32 == 0 && ldg % 8 == 0 && ldw0t % 8 == 0 && ldw0t >= N && lddw >= 16, "wide_tn_din: N must be a multiple of 32, at most 512");
  ALN_REQUIRE(((uintptr_t)g & 15) == 0 && ((uintptr_t)w0t & 15) == 0 && ((uintptr_t)geo & 15) == 0 && ((uintptr_t)d_fin & 7) == 0, "wide_tn_din: operands must be 16-byte aligned");
  if (M == 0) return 0;
  WideTNDin p{(const h16*)g, ldg, (const h16*)geo, G, (const h16*)w0t, ldw0t, M, N, 0, (float*)ws, (h16*)d_fin, found_inf};
  int slabs;
  wide_tn_din_split(M, p.slab, slabs);
  const size_t lds = (size_t)WTN_BM * WTD_PG * 2 + (size_t)WTN_BM * 24 * 2 + 2 * 8 * 64 * sizeof(float);
  static const bool lds_ok = hipFuncSetAttribute((const void*)k_wide_tn_din, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)) == hipSuccess;
  ALN_REQUIRE(lds_ok, "wide_tn_din: cannot reserve the LDS tile");
  hipLaunchKernelGGL(k_wide_tn_din, dim3((unsigned)slabs), dim3(256), lds, (hipStream_t)stream, p);
  ALN_CHECK_LAUNCH("wide_tn_din");
  const int64_t nk = (int64_t)N * 16;
  hipLaunchKernelGGL(k_wide_dw_reduce, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, slabs, N, 16, dw, lddw);
  ALN_CHECK_LAUNCH("wide_dw_reduce");
  return 0;
}

// row-major transpose of an fp16 matrix [R, C] -> [C, R] (W^T copies for the data gradients; weights only: tiny)
__global__ void k_transpose_h16(const h16* __restrict__ src, int R, int C, h16* __restrict__ dst) {
  __shared__ h16 t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int r = by + j, cidx = bx + threadIdx.x;
    t[j][threadIdx.x] = (r < R && cidx < C) ? src[(size_t)r * C + cidx] : (h16)0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int cidx = bx + j, r = by + threadIdx.x;
    if (cidx < C && r < R) dst[(size_t)cidx * R + r] = t[threadIdx.x][j];
  }
}
extern "C" int aln_transpose_f16(const void* src, int32_t R, int32_t C, void* dst, void* stream) {
  ALN_REQUIRE(src && dst && R > 0 && C > 0, "transpose_f16: bad arguments");
  hipLaunchKernelGGL(k_transpose_h16, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, (const h16*)src, R, C, (h16*)dst);
  ALN_CHECK_LAUNCH("transpose_f16");
  return 0;
}
