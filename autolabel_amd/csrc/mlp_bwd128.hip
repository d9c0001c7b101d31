// Fused backward WITH forward recompute of the 128-wide heads (sigma_net 48->128->128->16, color_net 32->128->128->16:
// autolabel/models.py:84-104), feature-sliced, one wave per SIMD.
//
// Round 3's kernel (k_mlp_bwd_recomp8, still used by the 64-wide heads) gave every chain wave 32 sample rows and ALL
// weights, read from LDS for every MFMA, plus four weight-gradient waves fed through LDS tiles: 57 KB of weights and 94 KB
// of tiles filled the LDS, the two roles shared a SIMD's matrix pipe, seven barriers per tile, 33 % MFMA busy.
// Here a block is four waves, one per SIMD (512 registers each), and wave w owns the hidden FEATURES [32w, 32w + 32) of
// both hidden layers for all 128 samples of a tile:
//   * its slice of every weight matrix (forward and transposed) is 80 registers and stays resident for the whole kernel:
//     no weight ever comes from LDS inside the tile loop (only W0^T, used once per tile for d_in, sits there);
//   * a layer is  Out[sample, n] = sum_k Act[sample, k] W[n, k]  with M = samples: the activations of ALL features come
//     from an LDS tile stored [feature][sample] through ds_read_b64_tr_b16 (lane = sample gets consecutive features),
//     the result lands with lane = feature, registers = samples;
//   * that C layout IS the A operand of the weight-gradient MFMAs (contraction over samples), so dA / h of the wave's own
//     slice feed dW straight from registers; only the other operand (all features of the layer input) is read from LDS;
//   * h2 never goes to LDS at all (its only consumers are the wave's own mask and dW_last), four barriers per tile
//     plus the tile load instead of seven.
// Sample order inside the k index of a weight-gradient MFMA is a pure summation order, so the C-register order
// ((r&3) + 8(r>>2) + 4*half) is used as is; the LDS operand follows it with two 8-byte reads (tr_frag_chained's order).
#include "mlp_shared.h"

namespace bwd128 {

constexpr int HID = 128, OUT = 16, KS = HID / 16, TR = 128, PH = 148 /* hid_pitch(128) */, PO = 16;
// pitch of the row-major x tile: IN + 8 keeps row-per-lane 16-byte reads conflict-free; the transposed reads of a 32-column block
// may run past IN into the next row (finite data, columns whose weight gradients are never flushed)
__host__ __device__ constexpr int px_pitch(int in) { return in + 8; }

struct Pk { u32x4 q[2]; };   // 16 packed halves of one 32-sample row block: C registers r = 0..15 of lane (feature, half)
__device__ inline h16x8 as_frag(u32x4 v) { return __builtin_bit_cast(h16x8, v); }
// Weight-gradient MFMA with its accumulator in the ACCUMULATOR half of the register file.  A one-wave-per-SIMD kernel owns 512
// registers, but only 256 of them are addressable by ordinary instructions: the 112 dW accumulators are touched by nothing but
// MFMAs until the final flush, so they live in AGPRs for the whole kernel, while the chain accumulators (converted, masked and
// packed by the VALU after every layer) stay in VGPRs (-amdgpu-mfma-vgpr-form: no v_accvgpr_read per element).  hipcc does not
// model an asm MFMA: the s_nop covers a VALU-written operand (cdna_hip_programming.md 5.7 item 2); back-to-back accumulation into
// the same registers needs no wait states, and the only other reader is the flush after the loop.
__device__ inline void mfma_acc(f32x16& acc, h16x8 a, h16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// A operand (lane = sample, 8 consecutive features in chained k order) from a [feature][sample] tile
__device__ inline h16x8 act_frag(const lds_h16* T, int rb, int ks, int lane) {
  return tr_frag_chained(PlainV<const lds_h16*>{T, PH}, 32 * rb, ks, lane);
}
// B operand of a weight-gradient MFMA (lane = feature of block ib, 8 samples in C-register order) from the same tile
__device__ inline h16x8 samp_frag(const lds_h16* T, int ib, int rb, int h, int lane) {
  const lds_h16* p0 = T + (32 * ib + (lane & 31)) * PH + 32 * rb + 16 * h + 4 * (lane >> 5);
  const lds_h16* p1 = p0 + 8;
  asm volatile("" : "+v"(p1));   // two ds_read_b64, never one ds_read2_b64 (mlp.hip: 8+ LDS cycles and 32-bank conflicts)
  union { struct { u32x2 a, b; } s; h16x8 v; } u;
  u.s.a = *(const LDS_VEC(u32x2)*)p0;
  u.s.b = *(const LDS_VEC(u32x2)*)p1;
  return u.v;
}
__device__ inline void relu_pack(const f32x16& acc, Pk& p) {
#pragma unroll
  for (int v = 0; v < 8; ++v) p.q[v >> 2][v & 3] = relu2(acc[2 * v], acc[2 * v + 1]);
}
// write the wave's 32-feature slice of one row block: lane (n, half) holds samples 8q + 4 half + 0..3 in registers 4q..4q+3
__device__ inline void write_slice(lds_h16* T, int wave, int rb, const Pk& p, int lane) {
  lds_h16* row = T + (32 * wave + (lane & 31)) * PH + 32 * rb + 4 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q) *(LDS_VEC(u32x2)*)(row + 8 * q) = (u32x2){p.q[q >> 1][2 * (q & 1)], p.q[q >> 1][2 * (q & 1) + 1]};
}

}  // namespace bwd128
using namespace bwd128;

// dev-only phase clock (scripts/dev/bench_mlp_bwd.py --phases, -DALN_PHASE_TIMING builds): shader-clock ticks between stamps, block 0 wave 0
#ifdef ALN_PHASE_TIMING
__device__ long long g_ph128[32];
#define PT_DECL long long pt_acc[32] = {0}; long long pt_last = clock64();
#define PT_STAMP(i) { long long pt_now = clock64(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; }
#define PT_FLUSH if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 32; ++i) g_ph128[i] += pt_acc[i]; }
#else
#define PT_DECL
#define PT_STAMP(i)
#define PT_FLUSH
#endif

template <int IN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_bwd128(const h16* __restrict__ wf_g, const h16* __restrict__ wb_g, const h16* __restrict__ x_g,
                  const h16* __restrict__ do_g, int rows, const int* __restrict__ rows_dev, h16* __restrict__ d_in,
                  float* __restrict__ dw_ws, int* __restrict__ found_inf) {
  constexpr int KS0 = IN / 16, IB = (IN + 31) / 32, PX = px_pitch(IN);
  constexpr int XCH = IN / 8, NXS = (TR * XCH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_h16* const t1 = (lds_h16*)smem;           // [128 features][PH]  h1   ([feature][sample])
  lds_h16* const t2 = t1 + HID * PH;            //                     dA2  (phase D: the d_in rows of the tile, staged for whole-row stores)
  lds_h16* const t3 = t2 + HID * PH;            //                     dA1
  lds_h16* const tXb = t3 + HID * PH;           // 2 x [128][PX]  x, row-major, double-buffered
  lds_h16* const tOb = tXb + 2 * TR * PX;       // 2 x [128][PO]  dL/dout, row-major
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (rows_dev) rows = min(rows, *rows_dev);
  const int ntiles = (rows + TR - 1) / TR;

  // ---- resident weight slices (fragment images of aln_mlp_repack: one 16-byte load per fragment and lane)
  const h16x8* const wf8 = (const h16x8*)wf_g;
  const h16x8* const wb8 = (const h16x8*)wb_g;
  h16x8 B0[KS0], B1[KS], B1T[KS], BL, W0T[IB][KS];
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks) B0[ks] = wf8[(size_t)(wave * KS0 + ks) * 64 + lane];            // W0[32w + n][natural k]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) B1[ks] = wf8[(size_t)(4 * KS0 + wave * KS + ks) * 64 + lane];    // W1[32w + n][chained k]
  BL = wb8[(size_t)wave * 64 + lane];                                                              // WL[o][32w + n]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) B1T[ks] = wb8[(size_t)(4 + wave * KS + ks) * 64 + lane];         // W1[chained o][32w + i]
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) W0T[ib][ks] = wb8[(size_t)(36 + ib * KS + ks) * 64 + lane];    // W0[chained o][32 ib + i]  (d_in: every wave, all of it)
  for (int i = threadIdx.x; i < (2 * TR * (PX + PO) + 64) / 8; i += 256) ((uint4*)tXb)[i] = make_uint4(0, 0, 0, 0);

  // ---- tile load: 16-byte chunks requested one tile ahead, parked in registers, stashed into the buffer of the NEXT tile
  h16x8 px[NXS], po;
  auto prefetch = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      const int i = threadIdx.x + 256 * q, r = i / XCH, k = i % XCH;
      if (i < TR * XCH && r0 + r < rows) px[q] = *(const h16x8*)(x_g + (size_t)(r0 + r) * IN + 8 * k);
    }
    const int r = threadIdx.x >> 1, k = threadIdx.x & 1;
    if (r0 + r < rows) po = *(const h16x8*)(do_g + (size_t)(r0 + r) * OUT + 8 * k);
  };
  auto stash = [&](int r0, lds_h16* tX, lds_h16* tO) __attribute__((always_inline)) {
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      const int i = threadIdx.x + 256 * q, r = i / XCH, k = i % XCH;
      if (i < TR * XCH) *(LDS_VEC(h16x8)*)(tX + r * PX + 8 * k) = (r0 + r < rows) ? px[q] : z;
    }
    const int r = threadIdx.x >> 1, k = threadIdx.x & 1;
    *(LDS_VEC(h16x8)*)(tO + r * PO + 8 * k) = (r0 + r < rows) ? po : z;
  };

  f32x16 dwl, dwm[4], dwf[IB];
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  dwl = zero16;
#pragma unroll
  for (int b = 0; b < 4; ++b) dwm[b] = zero16;
#pragma unroll
  for (int b = 0; b < IB; ++b) dwf[b] = zero16;
  h16x2 nanz = {0, 0};
#define SCHED_FENCE __builtin_amdgcn_sched_barrier(0)

  // h1 = relu(x W0^T), own slice -> t1 (first layer of a tile; runs one tile ahead, inside phase D of the tile before)
  auto layer0 = [&](const lds_h16* tX) __attribute__((always_inline)) {
    f32x16 acc[4];
    h16x8 a[2][4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) a[0][rb] = *(const LDS_VEC(h16x8)*)(tX + (32 * rb + c) * PX + 8 * hf);
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      if (ks + 1 < KS0) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) a[(ks + 1) & 1][rb] = *(const LDS_VEC(h16x8)*)(tX + (32 * rb + c) * PX + 16 * (ks + 1) + 8 * hf);
      }
      SCHED_FENCE;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(a[ks & 1][rb], B0[ks], ks == 0 ? zero16 : acc[rb]);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) { Pk hp; relu_pack(acc[rb], hp); write_slice(t1, wave, rb, hp, lane); }
  };

  // The weight loads must have LANDED before the loop is entered: otherwise hipcc's wait-count pass, merging the loop's two
  // entries, guards the first use of every weight register inside the loop with a counted vmcnt wait -- which, from the second
  // pass on, waits for the tile prefetch issued a moment earlier instead (measured: 2 000 ticks of exposed memory latency per tile).
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks) asm volatile("" : "+v"(B0[ks]));
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(B1[ks])); asm volatile("" : "+v"(B1T[ks])); }
  asm volatile("" : "+v"(BL));
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(W0T[ib][ks]));
  PT_DECL
  if ((int)blockIdx.x < ntiles) {
    prefetch(blockIdx.x * TR);
    __syncthreads();                       // zero fill done
    stash(blockIdx.x * TR, tXb, tOb);
    { const int nt = blockIdx.x + gridDim.x; if (nt < ntiles) prefetch(nt * TR); }
    __syncthreads();
    layer0(tXb);
    __syncthreads();
  }
  int cur = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const int r0 = tile * TR;
    const lds_h16* const tX = tXb + cur * (TR * PX);
    const lds_h16* const tO = tOb + cur * (TR * PO);
    lds_h16* const tXn = tXb + (cur ^ 1) * (TR * PX);
    lds_h16* const tOn = tOb + (cur ^ 1) * (TR * PO);
    const bool more = tile + (int)gridDim.x < ntiles;
    f32x16 acc[4];
    PT_STAMP(0)
    // ---- phase B: dH2 = dOut WL (while the first h1 fragments are on their way), h2 = relu(h1 W1^T) (registers only),
    //      dA2 = dH2 * relu'(h2) -> t2, dW_last
    Pk da2[4];
    {
      h16x8 a[2][4], ao[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) ao[rb] = *(const LDS_VEC(h16x8)*)(tO + (32 * rb + c) * PO + 8 * hf);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[0][rb] = act_frag(t1, rb, 0, lane);
      SCHED_FENCE;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(ao[rb], BL, zero16);
      Pk dh2[4];   // dH2 as packed halves (the conversion mask2 would do; the ReLU mask follows once h2 exists)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int v = 0; v < 8; ++v) dh2[rb].q[v >> 2][v & 3] = cvt_pk(acc[rb][2 * v], acc[rb][2 * v + 1]);
      h16x8 fo[8];   // dOut^T fragments of dW_last (A operand), requested during the layer
      PT_STAMP(20)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) a[(ks + 1) & 1][rb] = act_frag(t1, rb, ks + 1, lane);
        }
        if (ks >= 4) {
          fo[2 * (ks - 4)] = tr_frag_chained(PlainV<const lds_h16*>{tO, PO}, 0, 2 * (ks - 4), lane);
          fo[2 * (ks - 4) + 1] = tr_frag_chained(PlainV<const lds_h16*>{tO, PO}, 0, 2 * (ks - 4) + 1, lane);
        }
        SCHED_FENCE;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(a[ks & 1][rb], B1[ks], ks == 0 ? zero16 : acc[rb]);
      }
      PT_STAMP(21)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        Pk h2p;
        relu_pack(acc[rb], h2p);
        // dW_last[o][32w + n] += sum_s dOut[s][o] h2[s][n]
        mfma_acc(dwl, fo[2 * rb], as_frag(h2p.q[0]));
        mfma_acc(dwl, fo[2 * rb + 1], as_frag(h2p.q[1]));
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          uint32_t m;
          asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(h2p.q[v >> 2][v & 3]));
          asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(da2[rb].q[v >> 2][v & 3]) : "v"(dh2[rb].q[v >> 2][v & 3]), "v"(m));
        }
        write_slice(t2, wave, rb, da2[rb], lane);
      }
    }
    PT_STAMP(22)
    // first two k steps of dW_mid's LDS operand (h1, complete since the barrier at the top): in flight across the barrier
    h16x8 bm[3][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) { bm[0][ib] = samp_frag(t1, ib, 0, 0, lane); bm[1][ib] = samp_frag(t1, ib, 0, 1, lane); }
    PT_STAMP(1) __syncthreads(); PT_STAMP(2)   // B_b: t2 complete

    // ---- phase C: dH1 = dA2 W1 (own input slice), dA1 = dH1 * relu'(h1) -> t3 ; dW_mid beside it.
    // The weight-gradient MFMAs of the first two k steps run while the first dA2 fragments arrive, those of the last two
    // beside the mask / pack epilogue of the chain.
    Pk da1[4];
    {
      h16x8 a[2][4];
      u32x2 m[4][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[0][rb] = act_frag(t2, rb, 0, lane);
      auto dw = [&](int ks) __attribute__((always_inline)) {   // dW_mid[32w + n][i] += sum_s dA2[s][n] h1[s][i], k step = (row block ks >> 1, half ks & 1)
        if (ks + 2 < KS) {
#pragma unroll
          for (int ib = 0; ib < 4; ++ib) bm[(ks + 2) % 3][ib] = samp_frag(t1, ib, (ks + 2) >> 1, (ks + 2) & 1, lane);
        }
        SCHED_FENCE;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) mfma_acc(dwm[ib], as_frag(da2[ks >> 1].q[ks & 1]), bm[ks % 3][ib]);
      };
      auto chain = [&](int ks) __attribute__((always_inline)) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) a[(ks + 1) & 1][rb] = act_frag(t2, rb, ks + 1, lane);
        }
        if (ks == KS - 2) {   // h1 of the own slice, for the mask
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) m[rb][q] = *(const LDS_VEC(u32x2)*)(t1 + (32 * wave + c) * PH + 32 * rb + 4 * hf + 8 * q);
        }
        SCHED_FENCE;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(a[ks & 1][rb], B1T[ks], ks == 0 ? zero16 : acc[rb]);
      };
      PT_STAMP(10)
      dw(0); dw(1);
      PT_STAMP(11)
      chain(0); chain(1);
      PT_STAMP(12)
#pragma unroll
      for (int ks = 2; ks < KS - 2; ++ks) { dw(ks); chain(ks); }
      PT_STAMP(13)
      chain(KS - 2); chain(KS - 1);
      PT_STAMP(14)
      dw(KS - 2);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        if (rb == 2) dw(KS - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          da1[rb].q[q >> 1][2 * (q & 1)] = mask2(acc[rb][4 * q], acc[rb][4 * q + 1], m[rb][q].x);
          da1[rb].q[q >> 1][2 * (q & 1) + 1] = mask2(acc[rb][4 * q + 2], acc[rb][4 * q + 3], m[rb][q].y);
        }
        write_slice(t3, wave, rb, da1[rb], lane);
      }
    }
    PT_STAMP(15)
    // the next tile's x / dOut rows (requested a tile ago) go into the other buffer; then the tile after that is requested
#ifndef ALN_ABL_NOSTASH
    if (more) {
#ifdef ALN_PHASE_TIMING
      __builtin_amdgcn_s_waitcnt(0x0F70);
      PT_STAMP(17)
#endif
      stash((tile + gridDim.x) * TR, tXn, tOn);
      PT_STAMP(18)
      const int nt = tile + 2 * gridDim.x;
      if (nt < ntiles) prefetch(nt * TR);
    }
#endif
    PT_STAMP(16)
    // phase D's LDS operands that do not depend on t3: x^T fragments of the first two k steps
    h16x8 bx[3][IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) { bx[0][ib] = tr_frag_chained(PlainV<const lds_h16*>{tX, PX}, 32 * ib, 0, lane); bx[1][ib] = tr_frag_chained(PlainV<const lds_h16*>{tX, PX}, 32 * ib, 1, lane); }
    PT_STAMP(3) __syncthreads(); PT_STAMP(4)   // B_c: t3 complete, the next tile's x / dOut in place, t1 / t2 free

    // ---- phase D: d_in^T[i][s] = sum_k W0[k][i] dA1[s][k] for the wave's 32 samples ; dW_first beside it ; then the first layer
    //      of the NEXT tile (h1 -> t1) and this tile's d_in rows through t2 as whole rows
    {
      f32x16 o[IB];
      h16x8 b[2];
      b[0] = act_frag(t3, wave, 0, lane);
      auto dwf_ = [&](int ks) __attribute__((always_inline)) {   // dW_first[32w + n][i] += sum_s dA1[s][n] x[s][i]
        if (ks + 2 < KS) {
#pragma unroll
          for (int ib = 0; ib < IB; ++ib) bx[(ks + 2) % 3][ib] = tr_frag_chained(PlainV<const lds_h16*>{tX, PX}, 32 * ib, ks + 2, lane);
        }
        SCHED_FENCE;
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) mfma_acc(dwf[ib], as_frag(da1[ks >> 1].q[ks & 1]), bx[ks % 3][ib]);
      };
      auto din = [&](int ks) __attribute__((always_inline)) {
        if (ks + 1 < KS) b[(ks + 1) & 1] = act_frag(t3, wave, ks + 1, lane);
        SCHED_FENCE;
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) o[ib] = mfma16(W0T[ib][ks], b[ks & 1], ks == 0 ? zero16 : o[ib]);
      };
      PT_STAMP(25)
      dwf_(0); dwf_(1);
      din(0); din(1);
#pragma unroll
      for (int ks = 2; ks < KS - 2; ++ks) { dwf_(ks); din(ks); }
      din(KS - 2); din(KS - 1);
      dwf_(KS - 2); dwf_(KS - 1);
      PT_STAMP(26)
      // lane (sample c, half) holds features 32 ib + 8 q + 4 half + 0..3 of its row: through the wave's [32][IN] corner of t2, so the
      // rows leave as contiguous 16-byte pieces (one row per lane and store touched 64 lines per instruction: 800 ticks per tile)
      lds_h16* const stage = t2 + wave * (32 * IN);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * ib + 8 * q + 4 * hf;
          if (f < IN) {
            h16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (h16)o[ib][4 * q + r];
            nanz = nan_fold((h16x2){v[0], v[1]}, nan_fold((h16x2){v[2], v[3]}, nanz));
            *(LDS_VEC(h16x4)*)(stage + c * IN + f) = v;
          }
        }
    }
    PT_STAMP(27)
#ifndef ALN_ABL_NOL0
    if (more) layer0(tXn);
#endif
    PT_STAMP(28)
#ifndef ALN_ABL_NOSTORE
    if (d_in) {
      constexpr int PIECES = 32 * IN / 8;   // 16-byte pieces of the wave's 32 rows (contiguous in d_in)
      const lds_h16* const stage = t2 + wave * (32 * IN);
      const int row0 = r0 + 32 * wave;
#pragma unroll
      for (int i = 0; i < (PIECES + 63) / 64; ++i) {
        const int pc = lane + 64 * i;
        if (pc < PIECES && row0 + pc / (IN / 8) < rows) *(u32x4*)(d_in + (size_t)row0 * IN + 8 * pc) = *(const LDS_VEC(u32x4)*)(stage + 8 * pc);
      }
    }
#endif
    PT_STAMP(5) __syncthreads(); PT_STAMP(6)   // end of tile: t1 holds the next tile's h1; t2 / t3 / this tile's x buffer are free again
  }
  PT_STAMP(0)
  PT_FLUSH
  bool bad = nan_bad(nanz);
  if (dw_ws) {
    // per-block slab of partial sums, folded in a fixed order by k_dw_reduce(_all) (mlp.hip): layout = the fp32 master block
    constexpr size_t off1 = (size_t)IN * HID, off2 = off1 + (size_t)HID * HID, n_w = off2 + (size_t)HID * OUT;
    float* const slab = dw_ws + (size_t)blockIdx.x * n_w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * hf;   // C row of this register
      if (m < OUT) { bad |= !(fabsf(dwl[r]) <= 3.0e38f); slab[off2 + (size_t)m * HID + 32 * wave + c] = dwl[r]; }
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) { bad |= !(fabsf(dwm[ib][r]) <= 3.0e38f); slab[off1 + (size_t)(32 * wave + m) * HID + 32 * ib + c] = dwm[ib][r]; }
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
        if (32 * ib + c < IN) { bad |= !(fabsf(dwf[ib][r]) <= 3.0e38f); slab[(size_t)(32 * wave + m) * IN + 32 * ib + c] = dwf[ib][r]; }
    }
  }
  if (found_inf && __any(bad) && lane == 0) atomicOr(found_inf, 1);
}

template <int IN>
static int launch(const AlnMlpDesc* m, const h16* x, const h16* d_out, int rows, const int* rows_dev, void* d_in, float* ws, int g,
           int* found_inf, hipStream_t s) {
  constexpr int PX = px_pitch(IN);
  constexpr size_t lds = (2 * (size_t)TR * (PX + PO) + 64 + 3 * (size_t)HID * PH) * 2;   // + 64 halves: the transposed reads of the last row run past it
  static_assert(lds <= 160 * 1024, "LDS");
  hipFuncSetAttribute((const void*)k_mlp_bwd128<IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_mlp_bwd128<IN>), dim3(g), dim3(256), lds, s, (const h16*)m->wf, (const h16*)m->wb, x, d_out, rows, rows_dev,
                     (h16*)d_in, ws, found_inf);
  return 0;
}

#ifdef ALN_PHASE_TIMING
extern "C" int aln_debug_read_phases128(long long* host_out, int reset) {
  if (reset) { long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph128), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ph128), sizeof(long long) * 32);
}
#endif

// backward of a 128-wide two-hidden-layer head with a 16-wide output from plain x / dL/dout rows; `ws` = the block slabs
// (NULL: no weight gradients), g = aln_mlp_bwd_blocks.  Returns -3 when the shape has no instantiation.
int aln_launch_bwd128(const AlnMlpDesc* m, const void* x, const void* d_out, int rows, const int* rows_dev, void* d_in, float* ws,
                      int g, int* found_inf, hipStream_t s) {
  if (m->hidden != 128 || m->n_hidden != 2 || m->out_pad != 16) return -3;
  switch (m->in_pad) {
    case 32: return launch<32>(m, (const h16*)x, (const h16*)d_out, rows, rows_dev, d_in, ws, g, found_inf, s);
    case 48: return launch<48>(m, (const h16*)x, (const h16*)d_out, rows, rows_dev, d_in, ws, g, found_inf, s);
    case 64: return launch<64>(m, (const h16*)x, (const h16*)d_out, rows, rows_dev, d_in, ws, g, found_inf, s);
    default: return -3;
  }
}
