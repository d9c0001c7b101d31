// Forward of the 128-wide heads (density: enc -> 128 -> 128 -> 16, colour: 32 -> 128 -> 128 -> 16; autolabel/models.py:84-136,
// 175-207) over plain fp16 input rows: the training step's and the renderer's launches.
//
// k_mlp_fwd (mlp.hip) keeps the weight fragments in LDS and re-reads all of them for every 32-row tile -- 48 KB per tile and
// wave: at three waves per SIMD the LDS read stream is as long as the matrix work, and the pack / ReLU instructions between two
// layers of a tile have nothing to overlap with (22-43 % MFMA busy, profiles/r04_pmc_summary.json).  Here:
//   * ONE wave per SIMD, every fragment of all three layers resident in the AGPR half of its register file (52 fragments = 208
//     registers for the 48-wide input; the matrix instructions read their A operand from there directly); LDS only stages the
//     next input rows;
//   * the wave walks TWO 32-row tiles at a time, a stage apart: while the matrix pipe runs a layer of one tile, the vector ALU packs
//     the other tile's previous layer (fp32 -> fp16, ReLU) -- the two instruction streams are interleaved in program order, one
//     piece of vector work behind every matrix instruction, which is what a single in-order wave needs to keep both pipes busy;
//   * activations never leave the registers (the accumulator layout of one layer is the B-operand layout of the next: the chained
//     k-order of mlp.hip), the next pair's input rows are requested as soon as layer 0 has consumed the current ones.
// The matrix instructions are asm statements (hipcc would copy an AGPR-resident A operand to the VGPR half before every builtin):
// the wait states hipcc's hazard recogniser would insert are spelled out -- s_nop before the first use of a freshly packed B
// operand, a 20-state fence before the vector ALU reads an accumulator.  Same fragment image, same k-order as k_mlp_fwd: same bits.
#include "common.h"
#include "mlp_shared.h"

namespace {
__device__ __attribute__((always_inline)) inline void mfma_first(f32x16& acc, const h16x8& w, const h16x8& b) {   // acc = W b
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
}
__device__ __attribute__((always_inline)) inline void mfma_next(f32x16& acc, const h16x8& w, const h16x8& b) {    // acc += W b
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
// the vector ALU may read these accumulators after this point (8-pass matrix instructions: 11 wait states after the last one)
__device__ __attribute__((always_inline)) inline void acc_fence(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __attribute__((always_inline)) inline void acc_fence1(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a)); }

// word j (0..31) of the packed ReLU'd activations of a tile: features in the chained k-order (mlp.hip: relu_pack_store)
__device__ __attribute__((always_inline)) inline void pack_word(const f32x16 (&acc)[4], h16x8 (&p)[8], int j) {
  const int m = j >> 3, r = 2 * (j & 7);
  uint32_t w = relu2(acc[m][r], acc[m][r + 1]);
  // pinned HERE, behind the matrix instruction it was written after: hipcc would otherwise sink the conversion down to the first
  // use of the word -- the next layer's matrix instructions -- where nothing overlaps it (measured: the stage that should hide the
  // packing took 340 cycles, the one after it 830)
  asm volatile("" : "+v"(w));
  ((uint32_t*)&p[2 * m + (r >> 3)])[(r & 7) >> 1] = w;
}
}  // namespace

#ifdef ALN_PHASE_TIMING
__device__ long long g_f128_cycles[8];
extern "C" int aln_debug_read_fwd128(long long* host_out, int reset) {
  if (reset) { long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_f128_cycles), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_f128_cycles), sizeof(long long) * 8);
}
#define F_DECL long long f_acc[8] = {0}, f_last = clock64();
#define F_STAMP(i) { long long f_now = clock64(); f_acc[i] += f_now - f_last; f_last = f_now; }
#define F_FLUSH if (blockIdx.x == 0 && wave == 0 && lane == 0) for (int i = 0; i < 8; ++i) g_f128_cycles[i] += f_acc[i];
#else
#define F_DECL
#define F_STAMP(i)
#define F_FLUSH
#endif
// LAYOUT (AlnMlpDesc.x_tiled): 0 = row-major input rows, 1 = the tiled layout, 2 = PAIR PLANES (round 6: what aln_encode_fwd_planes writes --
// plane q = features (2 q, 2 q + 1) of every row, `pitch` words apart).  A COMPILE-TIME switch: as a run-time argument the address
// selects cost the row-major instantiations 60 us per training step (same-box A/B of the replayed step: 1.74 -> 1.80 ms).
// Planes: a request is still ONE 16-byte global_load_lds per lane, tile and k-step -- the lane fetches four consecutive ROWS of one plane
// (lane = 8 pos + row group; the eight planes of a k-step in the order 0 4 1 5 2 6 3 7, so that the words of the two halves of the wave sit
// 32 banks apart) -- and the read-back gathers the lane's four words (its row, planes 8 ks + 4 hf + 0..3) with two ds_read2_b32.
template <int KS0, int LAYOUT>   // k-steps of the input rows (in_pad / 16: 2 = colour head, 3 = density head)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_fwd128(const h16* __restrict__ wf, const h16* __restrict__ x, int rows, const int* __restrict__ rows_dev, h16* __restrict__ out,
                  float* __restrict__ sigma, long pitch) {
  constexpr bool TILED = LAYOUT == 1;
  constexpr int NB = 4, KS = 8, IN = 16 * KS0;
  const int lane = threadIdx.x & 63, hf = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (rows_dev) rows = min(rows, *rows_dev);
  // no live row (a batch whose compositing weights are all below the threshold): the clamped row index of the prefetch below would
  // be -1 and the requests would read up to 4 KB BEFORE the input buffer (a memory access fault when that page is unmapped)
  if (rows <= 0) return;
  const h16x8* frag = (const h16x8*)wf;
  constexpr int F1 = NB * KS0, FL = F1 + NB * KS;
  h16x8 W0[NB][KS0], W1[NB][KS], W2[KS];
#pragma unroll
  for (int m = 0; m < NB; ++m) {
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) W0[m][ks] = frag[(size_t)(m * KS0 + ks) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) W1[m][ks] = frag[(size_t)(F1 + m * KS + ks) * 64 + lane];
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) W2[ks] = frag[(size_t)(FL + ks) * 64 + lane];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the fragments have landed before the loop (hipcc merges loop-entry waits otherwise)

  const int ntiles = (rows + 31) / 32, npairs = (ntiles + 1) / 2;
  const int pstride = (int)gridDim.x * 4;
  int pair = (int)blockIdx.x * 4 + wave;
  // Input rows, TWO pairs ahead (a trip to HBM takes longer than one pair's 3 000 cycles of matrix work, and one wave per SIMD has
  // nobody else to hide it): fetched straight into LDS (global_load_lds: the data never passes through a register the compiler
  // could copy before it has landed, nothing is carried live across the loop edge), 16 bytes per lane in instruction order -- the
  // lane reads back its own slot.  Two staging buffers per wave, used alternately.  The read-back is an asm statement with its
  // own wait: hipcc would otherwise make every LDS read wait for ALL direct-to-LDS loads in flight, the younger request included.
  __shared__ __attribute__((aligned(16))) h16 stage[4][2][2 * KS0][64 * 8];
  auto request = [&](int p, int buf) __attribute__((always_inline)) {
    if constexpr (LAYOUT == 2) {
      const int pos = lane >> 3, rg = lane & 7, ql = (pos >> 1) + 4 * (pos & 1);
      const int last = ((rows - 1) >> 5) << 5;                                // first row of the last tile (rows % 32 == 0: the launcher checks)
      const int ta = min(p * 64, last), tb = min(p * 64 + 32, last);          // clamped: a tile beyond the end is never stored
      const uint32_t* const pl = (const uint32_t*)x + (size_t)ql * pitch + 4 * rg;
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl + (size_t)(8 * ks) * pitch + ta),
                                         (__attribute__((address_space(3))) void*)&stage[wave][buf][ks][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pl + (size_t)(8 * ks) * pitch + tb),
                                         (__attribute__((address_space(3))) void*)&stage[wave][buf][KS0 + ks][0], 16, 0, 0);
      }
      return;
    }
    const int ra = min(p * 64 + c, rows - 1), rb = min(p * 64 + 32 + c, rows - 1);   // clamped: a row beyond the end is never stored
    // row-major rows, or the tiled layout of AlnMlpDesc.x_tiled (piece 2 ks + hf of row r at 32 IN (r / 32) + 256 piece + 8 (r % 32)): the
    // pieces the lanes fetch are the same either way, only their addresses differ (tiled: a tile's piece is one contiguous 512 bytes)
    const h16* const pa = TILED ? x + (size_t)(ra >> 5) * (32 * IN) + 8 * (ra & 31) + 256 * hf : x + (size_t)ra * IN + 8 * hf;
    const h16* const pb = TILED ? x + (size_t)(rb >> 5) * (32 * IN) + 8 * (rb & 31) + 256 * hf : x + (size_t)rb * IN + 8 * hf;
    constexpr int kstep = TILED ? 512 : 16;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + kstep * ks),
                                       (__attribute__((address_space(3))) void*)&stage[wave][buf][ks][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + kstep * ks),
                                       (__attribute__((address_space(3))) void*)&stage[wave][buf][KS0 + ks][0], 16, 0, 0);
    }
  };
  h16x8 xA[KS0], xB[KS0];
  const uint32_t my_slot = (uint32_t)(size_t)(__attribute__((address_space(3))) h16*)&stage[wave][0][0][8 * lane];
  const uint32_t my_words = (uint32_t)(size_t)(__attribute__((address_space(3))) h16*)&stage[wave][0][0][0] + 128u * (uint32_t)hf + 4u * (uint32_t)c;
  auto arrived = [&](int buf) __attribute__((always_inline)) {
    // loads return in order: at most the 2 KS0 loads of the YOUNGER request may still be in flight (stores only make this wait
    // longer); every iteration issues exactly one request, so the count is the same everywhere
    if constexpr (LAYOUT == 2) {
      // slot s (1 KB) = [pos 0..7][32 rows] words; word j of the lane = plane 8 ks + 4 hf + j = pos 2 j + hf: 256 j + 128 hf + 4 c bytes in
      const uint32_t a0 = my_words + (uint32_t)buf * (2 * KS0 * 1024);
      u32x2 lo[2 * KS0], hi[2 * KS0];
      if constexpr (KS0 == 2)
        asm volatile("s_waitcnt vmcnt(4)\n\t"
                     "ds_read2_b32 %0, %8 offset1:64\n\tds_read2_b32 %1, %8 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %2, %9 offset1:64\n\tds_read2_b32 %3, %9 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %4, %10 offset1:64\n\tds_read2_b32 %5, %10 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %6, %11 offset1:64\n\tds_read2_b32 %7, %11 offset0:128 offset1:192\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
                     : "v"(a0), "v"(a0 + 1024u), "v"(a0 + 2048u), "v"(a0 + 3072u) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(6)\n\t"
                     "ds_read2_b32 %0, %12 offset1:64\n\tds_read2_b32 %1, %12 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %2, %13 offset1:64\n\tds_read2_b32 %3, %13 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %4, %14 offset1:64\n\tds_read2_b32 %5, %14 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %6, %15 offset1:64\n\tds_read2_b32 %7, %15 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %8, %16 offset1:64\n\tds_read2_b32 %9, %16 offset0:128 offset1:192\n\t"
                     "ds_read2_b32 %10, %17 offset1:64\n\tds_read2_b32 %11, %17 offset0:128 offset1:192\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3]),
                       "=&v"(lo[4]), "=&v"(hi[4]), "=&v"(lo[5]), "=&v"(hi[5])
                     : "v"(a0), "v"(a0 + 1024u), "v"(a0 + 2048u), "v"(a0 + 3072u), "v"(a0 + 4096u), "v"(a0 + 5120u) : "memory");
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) {
        xA[ks] = __builtin_bit_cast(h16x8, (u32x4){lo[ks][0], lo[ks][1], hi[ks][0], hi[ks][1]});
        xB[ks] = __builtin_bit_cast(h16x8, (u32x4){lo[KS0 + ks][0], lo[KS0 + ks][1], hi[KS0 + ks][0], hi[KS0 + ks][1]});
      }
      return;
    }
    const uint32_t at = my_slot + (uint32_t)buf * (2 * KS0 * 1024);
    if constexpr (KS0 == 2)
      asm volatile("s_waitcnt vmcnt(4)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                   "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(xA[0]), "=&v"(xA[1]), "=&v"(xB[0]), "=&v"(xB[1]) : "v"(at) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(6)\n\tds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                   "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(xA[0]), "=&v"(xA[1]), "=&v"(xA[2]), "=&v"(xB[0]), "=&v"(xB[1]), "=&v"(xB[2]) : "v"(at) : "memory");
  };
  request(min(pair, npairs - 1), 0);
  request(min(pair + pstride, npairs - 1), 1);
  int it = 0;
  f32x16 aA[NB], aB[NB], aC[NB];   // aC: layer 1 of tile A (its layer-0 results in aA are still being packed while it runs)
  h16x8 pA[KS], pB[KS];
  int rowB_prev = -1;   // tile B of the previous pair: its output leaves during layer 0 of this pair's tile A
  // output pieces: 0, 1 = the two 8-byte halves of the lane's part of the row, 2 = sigma
  auto out_piece = [&](const f32x16& o, int row, int q) __attribute__((always_inline)) {
    if (row >= 0 && row < rows) {
      if (q < 2) {
        h16x4 v; v[0] = (h16)o[4 * q]; v[1] = (h16)o[4 * q + 1]; v[2] = (h16)o[4 * q + 2]; v[3] = (h16)o[4 * q + 3];
        *(h16x4*)(out + (size_t)row * 16 + 8 * q + 4 * hf) = v;
      } else if (sigma && hf == 0) {
        // density head (models.py:175-188): sigma = trunc_exp(h0) of the fp16 output, written by the lane that holds feature 0
        sigma[row] = expf((float)(h16)o[0]);
      }
    }
  };
  F_DECL
  for (; pair < npairs; pair += pstride, ++it) {
    const int rowA = pair * 64 + c, rowB = rowA + 32;
    F_STAMP(7)
    arrived(it & 1);
    F_STAMP(0)
    // ---- stage 1: layer 0 of A  |  output of the previous pair's B
    if (rowB_prev >= 0) acc_fence1(aB[0]);
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (ks == 0) mfma_first(aA[m], W0[m][0], xA[0]); else mfma_next(aA[m], W0[m][ks], xA[ks]);
        if (ks * NB + m < 3) out_piece(aB[0], rowB_prev, ks * NB + m);
        __builtin_amdgcn_sched_barrier(0);
      }
    F_STAMP(1)
    // ---- stage 2: layer 0 of B  |  pack A, first half (pA[0..3]: what layer 1 needs for its first 16 matrix instructions)
    acc_fence(aA[0], aA[1], aA[2], aA[3]);
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (ks == 0) mfma_first(aB[m], W0[m][0], xB[0]); else mfma_next(aB[m], W0[m][ks], xB[ks]);
        constexpr int NS = NB * KS0;   // 16 words over NS slots
        const int i = ks * NB + m;
#pragma unroll
        for (int j = i * 16 / NS; j < (i + 1) * 16 / NS; ++j) pack_word(aA, pA, j);
        __builtin_amdgcn_sched_barrier(0);
      }
    // layer 0 has consumed this buffer: it takes the rows of the pair after the next (a pair beyond the end: clamped, never used)
    request(min(pair + 2 * pstride, npairs - 1), it & 1);
    F_STAMP(2)
    // ---- stage 3: layer 1 of A (into aC)  |  pack A, second half (pA[4..7], used from the 17th instruction on), pack B
    acc_fence(aB[0], aB[1], aB[2], aB[3]);
    asm volatile("s_nop 1" : "+v"(pA[3]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (ks == 0) mfma_first(aC[m], W1[m][0], pA[0]); else mfma_next(aC[m], W1[m][ks], pA[ks]);
        if (ks < 4) pack_word(aA, pA, 16 + ks * NB + m);
        pack_word(aB, pB, ks * NB + m);
        __builtin_amdgcn_sched_barrier(0);
      }
    F_STAMP(3)
    // ---- stage 4: layer 1 of B  |  pack A (layer 1)
    acc_fence(aC[0], aC[1], aC[2], aC[3]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int m = 0; m < NB; ++m) {
        if (ks == 0) mfma_first(aB[m], W1[m][0], pB[0]); else mfma_next(aB[m], W1[m][ks], pB[ks]);
        pack_word(aC, pA, ks * NB + m);     // (stage 3's matrix instructions have read pA long ago)
        __builtin_amdgcn_sched_barrier(0);
      }
    F_STAMP(4)
    // ---- stage 5: last layer of A  |  pack B (layer 1), words 0..23 (pB[0..5]; all of aB[0], which stage 6 overwrites)
    acc_fence(aB[0], aB[1], aB[2], aB[3]);
    asm volatile("s_nop 1" : "+v"(pA[7]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks == 0) mfma_first(aA[0], W2[0], pA[0]); else mfma_next(aA[0], W2[ks], pA[ks]);
#pragma unroll
      for (int j = 3 * ks; j < 3 * ks + 3; ++j) pack_word(aB, pB, j);
      __builtin_amdgcn_sched_barrier(0);
    }
    F_STAMP(5)
    // ---- stage 6: last layer of B  |  pack B words 24..31 (pB[6..7], from aB[3]), then the output of A
    acc_fence1(aA[0]);
    asm volatile("s_nop 1" : "+v"(pB[5]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks == 0) mfma_first(aB[0], W2[0], pB[0]); else mfma_next(aB[0], W2[ks], pB[ks]);
      if (ks < 4) { pack_word(aB, pB, 24 + 2 * ks); pack_word(aB, pB, 25 + 2 * ks); }
      else if (ks < 7) out_piece(aA[0], rowA, ks - 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    rowB_prev = rowB;
    F_STAMP(6)
  }
  F_FLUSH
  if (rowB_prev >= 0) {
    acc_fence1(aB[0]);
    out_piece(aB[0], rowB_prev, 0);
    out_piece(aB[0], rowB_prev, 1);
    out_piece(aB[0], rowB_prev, 2);
  }
}

// -3: shape not instantiated (the caller falls back to k_mlp_fwd)
int aln_launch_fwd128(const AlnMlpDesc* m, const void* x, int rows, const int* rows_dev, void* out, float* sigma, hipStream_t s) {
  if (m->hidden != 128 || m->n_hidden != 2 || m->out_pad != 16 || (m->in_pad != 32 && m->in_pad != 48)) return -3;
  const int pairs = (rows + 63) / 64, g = min(256, (pairs + 3) / 4);
  if (m->x_tiled == 2 && (rows % 32 != 0 || m->x_pitch < rows || m->x_pitch % 4 != 0 || ((uintptr_t)x & 15) != 0 || rows_dev)) {
    aln_set_error("mlp_fwd128: pair-plane input (x_tiled = 2) needs rows %% 32 == 0, x_pitch >= rows and a multiple of 4, a 16-byte aligned base");
    return -1;
  }
#define ALN_F128(KS0, T) hipLaunchKernelGGL((k_mlp_fwd128<KS0, T>), dim3(g), dim3(256), 0, s, (const h16*)m->wf, (const h16*)x, rows, rows_dev, (h16*)out, sigma, (long)m->x_pitch)
  if (m->in_pad == 32) { if (m->x_tiled == 2) ALN_F128(2, 2); else if (m->x_tiled) ALN_F128(2, 1); else ALN_F128(2, 0); }
  else { if (m->x_tiled == 2) ALN_F128(3, 2); else if (m->x_tiled) ALN_F128(3, 1); else ALN_F128(3, 0); }
#undef ALN_F128
  return 0;
}

