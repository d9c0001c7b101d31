// Pieces shared by the MLP translation units (mlp.hip, mlp_bwd128.hip): fragment bookkeeping of the fp16 weight images,
// packed-half helpers, LDS transpose reads.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) h16 lds_h16;   // LDS-typed element: ds_* with 32-bit addresses + immediate offsets
#define LDS_VEC(T) __attribute__((address_space(3))) T

// ---------------------------------------------------------------- LDS tile layouts
// Measured (scripts/dev/probe_lds_banks.hip, probe_mfma_rate.hip; SQ_LDS_BANK_CONFLICT = 61 % of the LDS cycles with rows
// padded to W + 8 halves): a ds_read_b64_tr_b16 fragment read ran 4-way bank-conflicted (its 32-lane group touches 4 rows x
// 16 dwords and the rows started 4 banks apart), row-per-lane 8-byte accesses 2-way, and the ds_read2_b64 the compiler
// forms from two adjacent 8-byte pieces costs 16 LDS cycles -- the MFMA phases were LDS-bound at 70-130 cycles per MFMA.
// Row pitches of the hidden-width tiles and matrices are therefore chosen by exhaustive search over the bank model
// (MI355X_MICROARCH.md, LDS): 148 halves for W = 128 and 68 for W = 64 make row-per-lane 8-byte reads and writes
// conflict-free and leave the transpose reads 2-way -- half the LDS cycles, with plain base + immediate addressing (an
// XOR/rotate swizzle reaches conflict-free transpose reads too but costs one address register per fragment: it spilled).
__host__ __device__ constexpr int hid_pitch(int hid) { return hid == 128 ? 148 : hid + 4; }
template <class TP>
struct PlainV {   // [rows][pitch] row-major
  TP p; int pitch;
  __device__ inline TP at(int row, int col) const { return p + row * pitch + col; }
};
template <class TP> __device__ inline PlainV<TP> plainv(TP p, int pitch) { return PlainV<TP>{p, pitch}; }

// ---------------------------------------------------------------- fragment bookkeeping
struct MlpLayers {
  int n;          // number of weight matrices (n_hidden + 1)
  int in_[3], out_[3];
  size_t w_off[3];  // offset of W_l in the fp32 master block
};
__host__ __device__ inline MlpLayers mlp_layers(int in_pad, int hid, int out_pad, int n_hidden) {
  MlpLayers L; L.n = n_hidden + 1;
  L.in_[0] = in_pad; L.out_[0] = hid;
  if (n_hidden == 2) { L.in_[1] = hid; L.out_[1] = hid; }
  L.in_[L.n - 1] = (L.n == 1) ? in_pad : hid; L.out_[L.n - 1] = out_pad;
  size_t o = 0;
  for (int l = 0; l < L.n; ++l) { L.w_off[l] = o; o += (size_t)L.in_[l] * L.out_[l]; }
  return L;
}
__host__ __device__ inline int ceil32(int x) { return (x + 31) / 32; }
// frag counts: forward layer l: ceil32(out) x in/16 ; backward layer l: ceil32(in) x out/16
__host__ __device__ inline size_t fwd_frag_off(const MlpLayers& L, int l) {
  size_t o = 0; for (int i = 0; i < l; ++i) o += (size_t)ceil32(L.out_[i]) * (L.in_[i] / 16); return o;
}
__host__ __device__ inline size_t bwd_frag_off(const MlpLayers& L, int l) {  // stored last layer first
  size_t o = 0; for (int i = L.n - 1; i > l; --i) o += (size_t)ceil32(L.in_[i]) * (L.out_[i] / 16); return o;
}
__host__ __device__ inline int kmap_natural(int ks, int hf, int j) { return 16 * ks + 8 * hf + j; }
__host__ __device__ inline int kmap_chained(int ks, int hf, int j) {
  return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hf + (j & 3);
}


__device__ inline f32x16 mfma16(h16x8 a, h16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ReLU + fp16 pack of a layer's C registers into next-layer B fragments; optional natural-layout store.
// Convert first (v_cvt_pk_f16_f32, RNE), then clamp the PACKED halves as signed 16-bit integers: every negative float
// (and -0) has the sign bit set, so max_i16(bits, 0) is ReLU on two values per instruction and never leaves a -0 behind
// (the backward masks below rely on "h > 0  <=>  bits != 0").
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t cvt_pk(float a, float b) {   // v_cvt_pk_f16_f32 (RNE)
  union { h16x2 h; uint32_t w; } u; u.h = __builtin_convertvector((f32x2){a, b}, h16x2);
  return u.w;
}
__device__ inline uint32_t relu2(float a, float b) {
  union { s16x2 i; uint32_t w; } u; u.w = cvt_pk(a, b);
  u.i = __builtin_elementwise_max(u.i, (s16x2){0, 0});
  return u.w;
}

// g * relu'(h) on two packed halves without compares or VCC traffic: h comes from relu2 (bits in [0, 0x7fff]), so
// min_u16(bits, 1) is 1 exactly where h > 0 and the 16-bit integer product with it keeps or clears the gradient bits.
// (inline asm: LLVM folds every C spelling of this back into compare + select)
__device__ inline uint32_t mask2(float g0, float g1, uint32_t h) {
  uint32_t g = cvt_pk(g0, g1), m;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(h));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(g) : "v"(g), "v"(m));
  return g;
}
// non-finite catcher on packed halves: x * 0 is NaN for inf / NaN, 0 otherwise
__device__ inline h16x2 nan_fold(h16x2 v, h16x2 acc) { return __builtin_elementwise_fma(v, (h16x2){0, 0}, acc); }
__device__ inline bool nan_bad(h16x2 acc) { return !((float)acc[0] == 0.f) || !((float)acc[1] == 0.f); }


typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4v* lds_s16x4_ptr;

template <class TV>
__device__ inline h16x8 tr_frag(TV t, int col0, int ks, int lane) {
  // operand fragment for mfma 32x32x16: lane (i = lane&31, hf = lane>>5) gets tile[16ks + 8hf + 0..7][col0 + i].
  // (the builtin lets the compiler count lgkmcnt itself, so several fragment reads stay in flight)
  const int hf = lane >> 5;
  const int row = 16 * ks + 8 * hf + ((lane & 15) >> 2);
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row, col));
  s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row + 4, col));
  union { struct { s16x4v l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}

// as tr_frag for an operand whose 8 samples per lane may be ANY 8 rows of the tile, as long as both operands of the MFMA use the
// same choice (a weight gradient sums over the samples: the order is free): k-step ks, half hf, read j take rows
// 32 (ks >> 1) + 8 q + 4 (ks & 1) + 2 hf + j, q = 0..3 -- the four rows of a 32-lane pass sit 8 rows apart, which tiles the 64 banks
// exactly for every pitch of 2 (mod 8) dwords (68 and 148 halves; consecutive rows collide two ways there).  Tiles of 32 n rows.
template <class TV>
__device__ inline h16x8 tr_frag_s8(TV t, int col0, int ks, int lane) {
  const int hf = lane >> 5;
  const int row = 32 * (ks >> 1) + 8 * ((lane & 15) >> 2) + 4 * (ks & 1) + 2 * hf;
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row, col));
  s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row + 1, col));
  union { struct { s16x4v l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}
template <class TV>
__device__ inline h16x8 tr_frag_chained(TV t, int col0, int ks, int lane) {
  // as tr_frag, but the 8 rows follow the chained k-order of the register chain: base + 8*(j>>2) + 4*hf + (j&3)
  const int hf = lane >> 5;
  const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row, col));
  s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)t.at(row + 8, col));
  union { struct { s16x4v l, h; } s; h16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}


// sum over the 32 lanes of the lane's half of the wave (every lane of the half gets the total): four DPP steps inside the rows of
// 16, one crossbar step between them
__device__ inline float half_sum32(float v) {
#define ALN_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  ALN_DPP_ADD(0xB1);    // quad_perm [1, 0, 3, 2]
  ALN_DPP_ADD(0x4E);    // quad_perm [2, 3, 0, 1]
  ALN_DPP_ADD(0x141);   // row_half_mirror
  ALN_DPP_ADD(0x140);   // row_mirror
#undef ALN_DPP_ADD
  return v + __shfl_xor(v, 16);
}
// mlp_bwd128.hip: recompute backward of the 128-wide heads (plain x / dL/dout rows); -3 = shape not instantiated
// dso != NULL: the dL/dout rows are assembled by the kernel's own loader from their three producers (the density head of the training step)
struct AlnDsoSrc { const float* d_h0; const void* d_semf_in; const void* d_color_in; const int* cidx_row; int G; };
int aln_launch_bwd128(const AlnMlpDesc* m, const void* x, const void* d_out, int rows, const int* rows_dev, void* d_in, float* ws,
                      int g, int* found_inf, hipStream_t s, const AlnDsoSrc* dso = nullptr);
// mlp_fwd128.hip: forward of the 128-wide heads over plain rows, weights resident in registers; -3 = shape not instantiated
int aln_launch_fwd128(const AlnMlpDesc* m, const void* x, int rows, const int* rows_dev, void* out, float* sigma, hipStream_t s);
