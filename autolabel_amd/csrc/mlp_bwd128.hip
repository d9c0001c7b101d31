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

// Physical row of feature f in the [feature][sample] tiles: inside every group of 32 the row index is rotated so that four
// CONSECUTIVE features sit eight rows apart.  The transposing read below fetches four consecutive features per 32-lane pass, 16
// dwords of each: at the 74-dword pitch consecutive rows start 10 banks apart and their windows collide two ways (PMC: 32-35 % of
// this kernel's LDS cycles were bank conflicts), rows eight apart start 16 banks apart and tile the 64 banks exactly.  The
// row-per-lane accesses (write_slice, samp_frag, the ReLU masks) touch the same 32 rows in another order: still conflict-free.
#ifdef ALN_PROW_OLD   // (dev builds only: A/B against the unrotated rows)
__device__ inline int prow(int f) { return f; }
#else
__device__ inline int prow(int f) { return (f & ~31) | ((f & 3) << 3) | ((f >> 2) & 7); }
#endif
// A operand (lane = sample, 8 consecutive features in chained k order) from a [feature][sample] tile
__device__ inline h16x8 act_frag(const lds_h16* T, int rb, int ks, int lane) {
  const int hf = lane >> 5;
  const int f = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);   // (tr_frag_chained's row; f + 8 stays in the group of 32)
  const int col = 32 * rb + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const lds_h16* p0 = T + prow(f) * PH + col;
  s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
  s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(T + prow(f + 8) * PH + col));   // (= p0 + 2 PH)
  union { struct { s16x4v l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}
// B operand of a weight-gradient MFMA (lane = feature of block ib, 8 samples in C-register order) from the same tile
__device__ inline h16x8 samp_frag(const lds_h16* T, int ib, int rb, int h, int lane) {
  const lds_h16* p0 = T + prow(32 * ib + (lane & 31)) * PH + 32 * rb + 16 * h + 4 * (lane >> 5);
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
  lds_h16* row = T + prow(32 * wave + (lane & 31)) * PH + 32 * rb + 4 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q) *(LDS_VEC(u32x2)*)(row + 8 * q) = (u32x2){p.q[q >> 1][2 * (q & 1)], p.q[q >> 1][2 * (q & 1) + 1]};
}

// Physical row of sample s in the row-major x tiles: inside every group of 16 the rows are rotated so that four CONSECUTIVE samples
// sit four rows apart.  The transposed reads for dW_first fetch four consecutive samples per 16-lane group, 16 columns of each: at
// the 28- / 20-dword pitches consecutive rows overlap two ways (profiles/r05_probe_lds_patterns.txt: half of those reads' cycles were
// conflicts), rows four apart start 48 / 16 banks apart and tile the 64 banks exactly.  Row-per-lane accesses see the same rows.
__device__ inline int xrow(int s_) { return (s_ & ~15) | ((s_ & 3) << 2) | ((s_ >> 2) & 3); }
// tr_frag_chained (mlp_shared.h) over such a tile
__device__ inline h16x8 tr_frag_x(const lds_h16* T, int pitch, int col0, int ks, int lane) {
  const int hf = lane >> 5;
  const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(T + xrow(row) * pitch + col));
  s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(T + xrow(row + 8) * pitch + col));
  union { struct { s16x4v l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
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

// DSO (round 6, the density head in the training step): the dL/dout rows do not exist in memory -- the loader builds row r itself from
// its three producers, exactly as k_assemble_dsigma_out (heads.hip) did in a pass of its own (28 us, 100 MB per step):
//   [ d_h0[r] | d_semf_in[r][0..G) + d_color_in[cidx[r]][16 .. 16 + G) ]     (autolabel/models.py:175-188: sigma = h0, geo_feat = h[1:])
// fp32 sums in the same order, one rounding to fp16, zeros from column G + 1 on; a value beyond the fp16 range raises found_inf.
struct DsoSrc { const float* d_h0; const h16* d_semf; const h16* d_color; const int* cidx; int G; };
template <int IN, int LAYOUT, bool DSO>   // LAYOUT = AlnMlpDesc.x_tiled: 0 row-major x rows, 1 tiled, 2 pair planes `pitch` words apart (compile-time: see mlp_fwd128.hip)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_bwd128(const h16* __restrict__ wf_g, const h16* __restrict__ wb_g, const h16* __restrict__ x_g,
                  const h16* __restrict__ do_g, int rows, const int* __restrict__ rows_dev, h16* __restrict__ d_in,
                  float* __restrict__ dw_ws, int* __restrict__ found_inf, long pitch, DsoSrc dso) {
  constexpr bool XT = LAYOUT == 1;
  constexpr int KS0 = IN / 16, IB = (IN + 31) / 32, PX = px_pitch(IN);
  constexpr int XCH = IN / 8, NXS = (TR * XCH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_h16* const t1 = (lds_h16*)smem;           // [128 features][PH]  h1   ([feature][sample])
  lds_h16* const t2 = t1 + HID * PH;            //                     dA2  (phase D: the d_in rows of the tile, staged for whole-row stores)
  lds_h16* const t3 = t2 + HID * PH;            //                     dA1
  lds_h16* const w0t = t3 + HID * PH;           // IB * 8 fragments of W0^T (wb image, layer 0): d_in's A operand, every wave reads all of it
  lds_h16* const tXb = w0t + IB * KS * 512;     // 2 x [128][PX]  x, row-major, double-buffered
  lds_h16* const tO = tXb + 2 * TR * PX;        // [128][PO]      dL/dout, row-major (dead after phase B: refilled in phase C)
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (rows_dev) rows = min(rows, *rows_dev);
  const int ntiles = (rows + TR - 1) / TR;

  // ---- resident weight slices (fragment images of aln_mlp_repack: one 16-byte load per fragment and lane)
  const h16x8* const wf8 = (const h16x8*)wf_g;
  const h16x8* const wb8 = (const h16x8*)wb_g;
  h16x8 B0[KS0], B1[KS], B1T[KS], BL;
#pragma unroll
  for (int ks = 0; ks < KS0; ++ks) B0[ks] = wf8[(size_t)(wave * KS0 + ks) * 64 + lane];            // W0[32w + n][natural k]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) B1[ks] = wf8[(size_t)(4 * KS0 + wave * KS + ks) * 64 + lane];    // W1[32w + n][chained k]
  BL = wb8[(size_t)wave * 64 + lane];                                                              // WL[o][32w + n]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) B1T[ks] = wb8[(size_t)(4 + wave * KS + ks) * 64 + lane];         // W1[chained o][32w + i]
  {
    const uint4* src = (const uint4*)(wb8 + (size_t)36 * 64);                                      // W0[chained o][32 ib + i]
    for (int i = threadIdx.x; i < IB * KS * 64; i += 256) ((uint4*)w0t)[i] = src[i];
    for (int i = threadIdx.x; i < (2 * TR * PX + TR * PO + 64) / 8; i += 256) ((uint4*)tXb)[i] = make_uint4(0, 0, 0, 0);
  }

  // ---- tile load: 16-byte chunks requested one tile ahead, parked in registers, stashed into the buffer of the NEXT tile
  h16x8 px[NXS], po;
  auto prefetch_x = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      // (pair planes: a wave takes 64 consecutive rows of ONE 16-byte chunk -- four planes, four coalesced 256-byte loads)
      const int i = threadIdx.x + 256 * q, r = LAYOUT == 2 ? i % TR : i / XCH, k = LAYOUT == 2 ? i / TR : i % XCH;
      // (x rows row-major, or in the tiled layout of AlnMlpDesc.x_tiled: chunk k of row m at 32 IN (m / 32) + 256 k + 8 (m % 32))
      if (i < TR * XCH && r0 + r < rows) {
        const int m = r0 + r;
        if constexpr (LAYOUT == 2) {
          const uint32_t* const pl = (const uint32_t*)x_g + (size_t)(4 * k) * pitch + m;
          px[q] = __builtin_bit_cast(h16x8, (u32x4){pl[0], pl[pitch], pl[2 * pitch], pl[3 * pitch]});
        } else
          px[q] = *(const h16x8*)(XT ? x_g + (size_t)(m >> 5) * (32 * IN) + 256 * k + 8 * (m & 31) : x_g + (size_t)m * IN + 8 * k);
      }
    }
  };
  h16x8 pc;                 // DSO: the colour head's chunk of the row (requested a phase after the row's compact index)
  int pci = -1; float ph0 = 0.f;
  h16x2 nanz = {0, 0};
  auto prefetch_o = [&](int r0) __attribute__((always_inline)) {
    const int r = threadIdx.x >> 1, k = threadIdx.x & 1;
    if constexpr (DSO) {
      pci = -1;
      if (r0 + r < rows) {
        const int m = r0 + r;
        po = *(const h16x8*)(dso.d_semf + (size_t)m * 16 + 8 * k);      // (chunk k of the semantic pair's d_semf_in row)
        pci = dso.cidx[m];
        if (k == 0) ph0 = dso.d_h0[m];
      }
    } else {
      if (r0 + r < rows) po = *(const h16x8*)(do_g + (size_t)(r0 + r) * OUT + 8 * k);
    }
  };
  auto prefetch_o2 = [&]() __attribute__((always_inline)) {   // DSO: the dependent load, one phase later (its index has arrived: no stall)
    if constexpr (DSO) {
      const int k = threadIdx.x & 1;
      if (pci >= 0) pc = *(const h16x8*)(dso.d_color + (size_t)pci * 32 + 16 + 8 * k);
    }
  };
  auto stash_x = [&](int r0, lds_h16* tX) __attribute__((always_inline)) {
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NXS; ++q) {
      const int i = threadIdx.x + 256 * q, r = LAYOUT == 2 ? i % TR : i / XCH, k = LAYOUT == 2 ? i / TR : i % XCH;
      if (i < TR * XCH) *(LDS_VEC(h16x8)*)(tX + xrow(r) * PX + 8 * k) = (r0 + r < rows) ? px[q] : z;
    }
  };
  auto stash_o = [&](int r0) __attribute__((always_inline)) {
    const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int r = threadIdx.x >> 1, k = threadIdx.x & 1;
    if constexpr (DSO) {
      // s[j] = geo_feat gradient 8 k + j; the output row is shifted by one column (column 0 = d_h0): this thread writes columns
      // 8 k .. 8 k + 7 = [d_h0 | left neighbour's s[7]], s[0..6]
      float sv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sv[j] = (float)po[j]; if (pci >= 0) sv[j] += (float)pc[j]; }
      const float left7 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[7]), 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true));
      h16x8 o;
      o[0] = k == 0 ? (h16)ph0 : (8 <= dso.G ? (h16)left7 : (h16)0.f);
#pragma unroll
      for (int j = 1; j < 8; ++j) o[j] = (8 * k + j <= dso.G) ? (h16)sv[j - 1] : (h16)0.f;
      const u32x4 ow = __builtin_bit_cast(u32x4, o);
      if (r0 + r < rows)
        nanz = nan_fold(__builtin_bit_cast(h16x2, ow[0]), nan_fold(__builtin_bit_cast(h16x2, ow[1]), nan_fold(__builtin_bit_cast(h16x2, ow[2]),
                        nan_fold(__builtin_bit_cast(h16x2, ow[3]), nanz))));
      *(LDS_VEC(h16x8)*)(tO + r * PO + 8 * k) = (r0 + r < rows) ? o : z;
    } else
      *(LDS_VEC(h16x8)*)(tO + r * PO + 8 * k) = (r0 + r < rows) ? po : z;
  };

  f32x16 dwl, dwm[4], dwf[IB];
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  dwl = zero16;
#pragma unroll
  for (int b = 0; b < 4; ++b) dwm[b] = zero16;
#pragma unroll
  for (int b = 0; b < IB; ++b) dwf[b] = zero16;
#define SCHED_FENCE __builtin_amdgcn_sched_barrier(0)

  // A fragments of the register chain come from an LDS tile through a ring of RING slots, requested DEPTH steps ahead: one wave
  // per SIMD has nothing but its own instruction stream to hide the ~300-tick LDS latency behind, and a step (two MFMAs of the
  // chain, two of the weight gradients) is 64-128 ticks long.
  constexpr int DEPTH = 4, RING = 5;
  constexpr int PIECES = 32 * IN / 8, NPC = (PIECES + 63) / 64;   // 16-byte pieces of a wave's 32 d_in rows (contiguous in d_in)
  u32x4 dst[NPC];                           // d_in rows of the PREVIOUS tile as whole pieces, stored early in the next phase B
  int dprev_row0 = -1;
  h16x8 l0a[4];                             // first-k-step x fragments of the next tile's first layer, requested before the barrier

  PT_DECL
  if ((int)blockIdx.x < ntiles) {
    prefetch_x(blockIdx.x * TR); prefetch_o(blockIdx.x * TR); prefetch_o2();
    __syncthreads();                       // zero fill done
    stash_x(blockIdx.x * TR, tXb); stash_o(blockIdx.x * TR);
    { const int nt = blockIdx.x + gridDim.x; if (nt < ntiles) { prefetch_x(nt * TR); prefetch_o(nt * TR); prefetch_o2(); } }
    __syncthreads();
    {   // first layer of the first tile (every later one runs inside phase D of the tile before)
      f32x16 acc[4];
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          acc[rb] = mfma16(*(const LDS_VEC(h16x8)*)(tXb + xrow(32 * rb + c) * PX + 16 * ks + 8 * hf), B0[ks], ks == 0 ? zero16 : acc[rb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) { Pk hp; relu_pack(acc[rb], hp); write_slice(t1, wave, rb, hp, lane); }
    }
    __syncthreads();
  }
  int cur = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const int r0 = tile * TR;
    const lds_h16* const tX = tXb + cur * (TR * PX);
    lds_h16* const tXn = tXb + (cur ^ 1) * (TR * PX);
    const bool more = tile + (int)gridDim.x < ntiles;
    const int nt2 = tile + 2 * gridDim.x;
    f32x16 acc[4];
    PT_STAMP(0)
    // =====================================================================================================================
    // phase B: h2 = relu(h1 W1^T) (registers only), dH2 = dOut WL, dA2 = dH2 * relu'(h2) -> t2, dW_last.  Beside it: the previous
    //          tile's d_in rows leave, the next tile's x rows go into the other buffer.
    // The chain runs row-block PAIR by pair (two independent accumulators per step), so the epilogue of the first pair (last
    // layer's backward, ReLU, mask, pack, dW_last) sits between the MFMAs of the second.
    Pk da2[4];
    {
      h16x8 a[RING][2], ao[2], fo[2][2];
      f32x16 lacc[2];
      auto req = [&](int s) __attribute__((always_inline)) {   // step s = (pair s >> 3, k step s & 7)
        a[s % RING][0] = act_frag(t1, 2 * (s >> 3), s & 7, lane);
        a[s % RING][1] = act_frag(t1, 2 * (s >> 3) + 1, s & 7, lane);
      };
      auto prep = [&](int rb) __attribute__((always_inline)) {   // dL/dout rows of the block (A operand of the last layer's backward) and their transposes (dW_last)
        ao[rb & 1] = *(const LDS_VEC(h16x8)*)(tO + (32 * rb + c) * PO + 8 * hf);
        fo[rb & 1][0] = tr_frag_chained(PlainV<const lds_h16*>{tO, PO}, 0, 2 * rb, lane);
        fo[rb & 1][1] = tr_frag_chained(PlainV<const lds_h16*>{tO, PO}, 0, 2 * rb + 1, lane);
      };
      auto last = [&](int rb) __attribute__((always_inline)) { lacc[rb & 1] = mfma16(ao[rb & 1], BL, zero16); };
      auto epilogue = [&](int rb) __attribute__((always_inline)) {
        Pk h2p;
        relu_pack(acc[rb], h2p);
        // dW_last[o][32w + n] += sum_s dOut[s][o] h2[s][n]
        mfma_acc(dwl, fo[rb & 1][0], as_frag(h2p.q[0]));
        mfma_acc(dwl, fo[rb & 1][1], as_frag(h2p.q[1]));
#pragma unroll
        for (int v = 0; v < 8; ++v) da2[rb].q[v >> 2][v & 3] = mask2(lacc[rb & 1][2 * v], lacc[rb & 1][2 * v + 1], h2p.q[v >> 2][v & 3]);
        write_slice(t2, wave, rb, da2[rb], lane);
      };
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) req(s);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s + DEPTH < 16) req(s + DEPTH);
        if (s == 2 && dprev_row0 >= 0 && d_in) {
#pragma unroll
          for (int i = 0; i < NPC; ++i) {
            const int pc = lane + 64 * i;
            if (pc < PIECES && dprev_row0 + pc / (IN / 8) < rows) *(u32x4*)(d_in + (size_t)dprev_row0 * IN + 8 * pc) = dst[i];
          }
        }
        if (s == 5 && more) stash_x((tile + gridDim.x) * TR, tXn);
        if (s == 6 && more && nt2 < ntiles) prefetch_x(nt2 * TR);
        if (s == 8) prep(0);
        if (s == 10) prep(1);
        if (s == 13) prep(2);
        SCHED_FENCE;
        const int p = s >> 3, ks = s & 7;
        acc[2 * p] = mfma16(a[s % RING][0], B1[ks], ks == 0 ? zero16 : acc[2 * p]);
        acc[2 * p + 1] = mfma16(a[s % RING][1], B1[ks], ks == 0 ? zero16 : acc[2 * p + 1]);
        if (s == 9) last(0);
        if (s == 10) epilogue(0);
        if (s == 11) last(1);
        if (s == 12) epilogue(1);
        if (s == 14) last(2);
      }
      prep(3);
      epilogue(2);
      last(3);
      epilogue(3);
    }
    // first two k steps of dW_mid's LDS operand (h1, complete since the barrier at the top): requested before the barrier
    h16x8 bm[3][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) { bm[0][ib] = samp_frag(t1, ib, 0, 0, lane); bm[1][ib] = samp_frag(t1, ib, 0, 1, lane); }
    PT_STAMP(1) __syncthreads(); PT_STAMP(2)   // B_b: t2 complete, the next tile's x in place, dOut dead

    // =====================================================================================================================
    // phase C: dH1 = dA2 W1 (own input slice), dA1 = dH1 * relu'(h1) -> t3 ; dW_mid beside it: its first two k steps while the first
    //          dA2 fragments arrive, its last beside the mask / pack epilogue of the second pair.  The next tile's dOut rows go in.
    Pk da1[4];
    {
      h16x8 a[RING][2];
      u32x2 m[2][4];
      auto req = [&](int s) __attribute__((always_inline)) {
        a[s % RING][0] = act_frag(t2, 2 * (s >> 3), s & 7, lane);
        a[s % RING][1] = act_frag(t2, 2 * (s >> 3) + 1, s & 7, lane);
      };
      auto dw = [&](int kd, int ib0, int n) __attribute__((always_inline)) {   // dW_mid[32w + n][i] += sum_s dA2[s][n] h1[s][i], k step kd = (row block, half)
#pragma unroll
        for (int ib = ib0; ib < ib0 + n; ++ib) mfma_acc(dwm[ib], as_frag(da2[kd >> 1].q[kd & 1]), bm[kd % 3][ib]);
      };
      auto req_bm = [&](int kd) __attribute__((always_inline)) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) bm[kd % 3][ib] = samp_frag(t1, ib, kd >> 1, kd & 1, lane);
      };
      auto req_m = [&](int rb, int j) __attribute__((always_inline)) {   // h1 of the own slice, for the mask
#pragma unroll
        for (int q = 0; q < 4; ++q) m[j][q] = *(const LDS_VEC(u32x2)*)(t1 + prow(32 * wave + c) * PH + 32 * rb + 4 * hf + 8 * q);
      };
      auto epilogue = [&](int rb, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          da1[rb].q[q >> 1][2 * (q & 1)] = mask2(acc[rb][4 * q], acc[rb][4 * q + 1], m[j][q].x);
          da1[rb].q[q >> 1][2 * (q & 1) + 1] = mask2(acc[rb][4 * q + 2], acc[rb][4 * q + 3], m[j][q].y);
        }
        write_slice(t3, wave, rb, da1[rb], lane);
      };
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) req(s);
      req_bm(2);
      SCHED_FENCE;
      dw(0, 0, 4);
      req_bm(3);
      SCHED_FENCE;
      dw(1, 0, 4);
      // k steps 2..6 of dW_mid: two MFMAs beside every step of the chain's first ten.  LDS requests are issued one or two at a time
      // between the MFMAs (a burst of eight against a full queue stalls the wave -- and every MFMA behind it -- until slots free up)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int p = s >> 3, ks = s & 7;
        if (s + DEPTH < 16) a[(s + DEPTH) % RING][0] = act_frag(t2, 2 * ((s + DEPTH) >> 3), (s + DEPTH) & 7, lane);
        if (s == 3 && more) stash_o((tile + gridDim.x) * TR);
        if (s == 4) { req_m(0, 0); if (more && nt2 < ntiles) prefetch_o(nt2 * TR); }
        if (s == 5) req_m(1, 1);
        SCHED_FENCE;
        acc[2 * p] = mfma16(a[s % RING][0], B1T[ks], ks == 0 ? zero16 : acc[2 * p]);
        SCHED_FENCE;
        if (s + DEPTH < 16) a[(s + DEPTH) % RING][1] = act_frag(t2, 2 * ((s + DEPTH) >> 3) + 1, (s + DEPTH) & 7, lane);
        SCHED_FENCE;
        acc[2 * p + 1] = mfma16(a[s % RING][1], B1T[ks], ks == 0 ? zero16 : acc[2 * p + 1]);
        SCHED_FENCE;
        if (s < 10) {
          const int kd = 2 + s / 2, ib0 = 2 * (s & 1), kn = kd + 2;   // operands of k step kd + 2 are requested beside k step kd
          if (kn < KS) bm[kn % 3][ib0] = samp_frag(t1, ib0, kn >> 1, kn & 1, lane);
          SCHED_FENCE;
          dw(kd, ib0, 1);
          SCHED_FENCE;
          if (kn < KS) bm[kn % 3][ib0 + 1] = samp_frag(t1, ib0 + 1, kn >> 1, kn & 1, lane);
          SCHED_FENCE;
          dw(kd, ib0 + 1, 1);
        }
        if (s == 10) { epilogue(0, 0); req_m(2, 0); }
        if (s == 12) { epilogue(1, 1); req_m(3, 1); }
      }
      dw(7, 0, 2);
      epilogue(2, 0);
      dw(7, 2, 2);
      epilogue(3, 1);
    }
    // phase D's LDS operands that do not depend on t3: x^T fragments of the first two k steps, the next tile's first x fragments
    h16x8 bx[3][IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) { bx[0][ib] = tr_frag_x(tX, PX, 32 * ib, 0, lane); bx[1][ib] = tr_frag_x(tX, PX, 32 * ib, 1, lane); }
    if (more) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) l0a[rb] = *(const LDS_VEC(h16x8)*)(tXn + xrow(32 * rb + c) * PX + 8 * hf);
    }
    PT_STAMP(3) __syncthreads(); PT_STAMP(4)   // B_c: t3 complete, t1 / t2 free

    // =====================================================================================================================
    // phase D: d_in^T[i][s] = sum_k W0[k][i] dA1[s][k] for the wave's 32 samples ; dW_first beside it ; and the first layer of the
    //          NEXT tile (h1 -> t1), whose MFMAs open the phase while the dA1 fragments arrive.
    {
      f32x16 o[IB];
      h16x8 b[RING], la[2][4], wa[2][IB];
      auto req = [&](int ks) __attribute__((always_inline)) { b[ks % RING] = act_frag(t3, wave, ks, lane); };
      auto req_w = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) wa[ks & 1][ib] = *(const LDS_VEC(h16x8)*)(w0t + ((ib * KS + ks) * 64 + lane) * 8);
      };
      auto dwf_ = [&](int ks) __attribute__((always_inline)) {   // dW_first[32w + n][i] += sum_s dA1[s][n] x[s][i]
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) mfma_acc(dwf[ib], as_frag(da1[ks >> 1].q[ks & 1]), bx[ks % 3][ib]);
      };
      auto req_bx = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) bx[ks % 3][ib] = tr_frag_x(tX, PX, 32 * ib, ks, lane);
      };
#pragma unroll
      for (int ks = 0; ks < DEPTH; ++ks) req(ks);
      req_w(0);
      if (more && nt2 < ntiles) prefetch_o2();   // (DSO: the colour chunk of the tile whose first loads went out in phase C)
      if (more) {
        if (KS0 > 1) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) la[1][rb] = *(const LDS_VEC(h16x8)*)(tXn + xrow(32 * rb + c) * PX + 16 + 8 * hf);
        }
        SCHED_FENCE;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(l0a[rb], B0[0], zero16);
      }
      req_bx(2);
      SCHED_FENCE;
      dwf_(0);
      if (more) {
#pragma unroll
        for (int ks = 1; ks < KS0; ++ks) {
          if (ks + 1 < KS0) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) la[(ks + 1) & 1][rb] = *(const LDS_VEC(h16x8)*)(tXn + xrow(32 * rb + c) * PX + 16 * (ks + 1) + 8 * hf);
          }
          SCHED_FENCE;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[rb] = mfma16(la[ks & 1][rb], B0[ks], acc[rb]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + DEPTH < KS) req(ks + DEPTH);
        if (ks + 1 < KS) req_w(ks + 1);
        if (ks + 3 < KS) req_bx(ks + 3);
        SCHED_FENCE;
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) o[ib] = mfma16(wa[ks & 1][ib], b[ks % RING], ks == 0 ? zero16 : o[ib]);
        if (ks + 1 < KS) dwf_(ks + 1);
        if (more && ks >= 1 && ks <= 4) { Pk hp; relu_pack(acc[ks - 1], hp); write_slice(t1, wave, ks - 1, hp, lane); }
      }
      // lane (sample c, half) holds features 32 ib + 8 q + 4 half + 0..3 of its row: through the wave's [32][IN] corner of t2 (free
      // since B_c), so the rows leave as contiguous 16-byte pieces (one row per lane and store touched 64 lines per instruction:
      // 800 ticks per tile); the pieces are read back here and stored early in the next tile's phase B
      // (pitch IN + 4 halves: at pitch IN the 32 rows of an 8-byte store fall onto 8 bank pairs -- four ways, 64 ticks per store)
      constexpr int PS = IN + 4;
      lds_h16* const stage = t2 + wave * (32 * PS);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * ib + 8 * q + 4 * hf;
          if (f < IN) {
            const u32x2 v = {cvt_pk(o[ib][4 * q], o[ib][4 * q + 1]), cvt_pk(o[ib][4 * q + 2], o[ib][4 * q + 3])};
            nanz = nan_fold(__builtin_bit_cast(h16x2, v.x), nan_fold(__builtin_bit_cast(h16x2, v.y), nanz));
            *(LDS_VEC(u32x2)*)(stage + c * PS + f) = v;
          }
        }
      SCHED_FENCE;
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int pc = lane + 64 * i;
        if (pc < PIECES) {   // piece pc = 16 bytes of row pc / (IN / 8): two 8-byte reads (the padded rows are 8-byte aligned only)
          const lds_h16* src = stage + (pc / (IN / 8)) * PS + 8 * (pc % (IN / 8));
          const u32x2 a = *(const LDS_VEC(u32x2)*)src, b = *(const LDS_VEC(u32x2)*)(src + 4);
          dst[i] = (u32x4){a.x, a.y, b.x, b.y};
        }
      }
      dprev_row0 = r0 + 32 * wave;
    }
    PT_STAMP(5) __syncthreads(); PT_STAMP(6)   // end of tile: t1 holds the next tile's h1; t2 / t3 / this tile's x buffer are free again
  }
  if (dprev_row0 >= 0 && d_in) {   // the last tile's d_in rows
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int pc = lane + 64 * i;
      if (pc < PIECES && dprev_row0 + pc / (IN / 8) < rows) *(u32x4*)(d_in + (size_t)dprev_row0 * IN + 8 * pc) = dst[i];
    }
  }
  PT_STAMP(0)
  PT_FLUSH
  // (hipcc does not know the asm statements are MFMAs: the flush below must not read an accumulator before its last pass has landed)
  asm volatile("s_nop 15" : "+a"(dwl), "+a"(dwm[0]), "+a"(dwm[1]), "+a"(dwm[2]), "+a"(dwm[3]), "+a"(dwf[0]));
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
           int* found_inf, hipStream_t s, const DsoSrc* dso) {
  constexpr int PX = px_pitch(IN);
  constexpr int IB = (IN + 31) / 32;
  constexpr size_t lds = (2 * (size_t)TR * PX + (size_t)TR * PO + 64 + 3 * (size_t)HID * PH + (size_t)IB * KS * 512) * 2;   // + 64 halves: the transposed reads of the last row run past it
  static_assert(lds <= 160 * 1024, "LDS");
#define ALN_B128(T, D)                                                                                                                      \
  do {                                                                                                                                      \
    hipFuncSetAttribute((const void*)k_mlp_bwd128<IN, T, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
    hipLaunchKernelGGL((k_mlp_bwd128<IN, T, D>), dim3(g), dim3(256), lds, s, (const h16*)m->wf, (const h16*)m->wb, x, d_out, rows, rows_dev, \
                       (h16*)d_in, ws, found_inf, (long)m->x_pitch, dso ? *dso : DsoSrc{});                                                 \
  } while (0)
  if (m->x_tiled == 2 && (m->x_pitch < rows || ((uintptr_t)x & 3) != 0)) { aln_set_error("mlp_bwd128: pair-plane input (x_tiled = 2) needs x_pitch >= rows"); return -1; }
  if (dso) {
    if constexpr (IN == 48) { if (m->x_tiled == 2) ALN_B128(2, true); else if (m->x_tiled) ALN_B128(1, true); else ALN_B128(0, true); }
    else return -3;
  } else if (m->x_tiled == 2) ALN_B128(2, false);
  else if (m->x_tiled) ALN_B128(1, false);
  else ALN_B128(0, false);
#undef ALN_B128
  return 0;
}

#ifdef ALN_PHASE_TIMING
extern "C" int aln_debug_read_phases128(long long* host_out, int reset) {
  if (reset) { long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph128), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ph128), sizeof(long long) * 32);
}
#endif

// backward of a 128-wide two-hidden-layer head with a 16-wide output from plain x / dL/dout rows; `ws` = the block slabs
// (NULL: no weight gradients), g = aln_mlp_bwd_blocks.  Returns -3 when the shape has no instantiation (a 64-wide input --
// the 'freq' encoding -- does not fit the LDS budget with a double-buffered x tile: it stays on k_mlp_bwd_recomp8).
int aln_launch_bwd128(const AlnMlpDesc* m, const void* x, const void* d_out, int rows, const int* rows_dev, void* d_in, float* ws,
                      int g, int* found_inf, hipStream_t s, const AlnDsoSrc* dso_in) {
  if (m->hidden != 128 || m->n_hidden != 2 || m->out_pad != 16) return -3;
  DsoSrc dso_v, *dso = nullptr;
  if (dso_in) { dso_v = DsoSrc{dso_in->d_h0, (const h16*)dso_in->d_semf_in, (const h16*)dso_in->d_color_in, dso_in->cidx_row, dso_in->G}; dso = &dso_v; }
  switch (m->in_pad) {
    case 32: return launch<32>(m, (const h16*)x, (const h16*)d_out, rows, rows_dev, d_in, ws, g, found_inf, s, dso);
    case 48: return launch<48>(m, (const h16*)x, (const h16*)d_out, rows, rows_dev, d_in, ws, g, found_inf, s, dso);
    default: return -3;
  }
}
