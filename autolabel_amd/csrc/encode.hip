// Frequency + multiresolution hash-grid encoding, forward and backward.
// Replaces tcnn Encoding{Frequency,Grid} called at autolabel/models.py:51-59 (HGFreqEncoder.forward),
// :25-27 (FreqEncoder.forward).  Spec: oracle/nerf_oracle.py (grid_corner_indices, hashgrid_encode,
// freq_encode).  Index arithmetic is bit-exact to the oracle (unfused fp32 pos, uint32 hash).
//
// HBM-bound kernel.  Work mapping: a 256-thread block owns a tile of 64 consecutive rows (samples of
// one ray are consecutive, so a wave's 64 lanes walk along a ray: coarse levels hit the same few cache
// lines, fine levels are random 4-byte gathers served by L2/Infinity Cache -- the fp16 table is
// 28.5 MB).  Wave w handles levels w, w+4, ...; results are staged in LDS and the tile is written as
// ONE contiguous span (64 rows * enc_pad halves) with 16-byte stores.
#include "common.h"
#include <math.h>

#define ENC_TILE 64
#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

struct EncParams {
  AlnEncDesc e;
  const uint32_t* table;  // fp16x2 per entry
  const float* rays_o; const float* rays_d; const float* z; const float* xyz;
  int rows, rays_stride;
  int level_lo, level_hi;   // backward only: levels [level_lo, level_hi) of this launch
};

__device__ inline void row_position(const EncParams& p, int row, float* x) {
  if (p.xyz) {
    x[0] = p.xyz[3 * (size_t)row]; x[1] = p.xyz[3 * (size_t)row + 1]; x[2] = p.xyz[3 * (size_t)row + 2];
  } else {
    int ray = row / p.rays_stride;
    aln_sample_xyz(p.rays_o + 3 * (size_t)ray, p.rays_d + 3 * (size_t)ray, p.z[row], p.e.bound, x);
  }
}

__device__ inline void normalize_pos(const float* x, float bound, bool clip, float* xn) {
  float two_b = 2.0f * bound;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = __fdiv_rn(__fadd_rn(x[k], bound), two_b);
    xn[k] = clip ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
  }
}

// level-local corner indices + trilinear weights (oracle: grid_corner_indices)
__device__ inline void grid_corners(const AlnGridDesc& g, int l, const float* xn, uint32_t* idx, float* w, uint32_t* base = nullptr) {
  float scale = g.scale[l];
  uint32_t res = g.res[l], size = g.size[l];
  bool dense = g.dense[l] != 0;
  uint32_t gi[3]; float fr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pos = __fadd_rn(__fmul_rn(xn[k], scale), 0.5f);
    float fl = floorf(pos);
    gi[k] = (uint32_t)(int)fl;
    fr[k] = __fsub_rn(pos, fl);
  }
  if (base) { base[0] = gi[0]; base[1] = gi[1]; base[2] = gi[2]; }
  // index terms per axis: the +1 corner is one add away from the base corner (uint32 wrap-around keeps (g + 1) * P exact),
  // and "% size" is a mask for the hashed levels (size = 2^k) and one conditional subtract for the dense ones
  // (index <= res + res^2 + res^3 < 2 * size): 2 integer multiplies per level instead of 16 and no division
  const uint32_t ty = dense ? res : PRIME_Y, tz = dense ? res * res : PRIME_Z;
  const uint32_t ay[2] = {gi[1] * ty, gi[1] * ty + ty}, az[2] = {gi[2] * tz, gi[2] * tz + tz};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float ww = 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ww = (c & (1 << k)) ? __fmul_rn(ww, fr[k]) : __fmul_rn(ww, __fsub_rn(1.0f, fr[k]));
    const uint32_t gx = gi[0] + (uint32_t)(c & 1);
    uint32_t i;
    if (dense) { i = gx + ay[(c >> 1) & 1] + az[c >> 2]; i = i >= size ? i - size : i; }
    else i = (gx ^ ay[(c >> 1) & 1] ^ az[c >> 2]) & (size - 1u);
    idx[c] = i;
    w[c] = ww;
  }
}

// trilinear weights + base cell only (forward: the indices are computed by the lanes that issue the loads)
__device__ inline void grid_weights(const AlnGridDesc& g, int l, const float* xn, float* w, uint32_t* base) {
  float scale = g.scale[l];
  float fr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pos = __fadd_rn(__fmul_rn(xn[k], scale), 0.5f);
    float fl = floorf(pos);
    base[k] = (uint32_t)(int)fl;
    fr[k] = __fsub_rn(pos, fl);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float ww = 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ww = (c & (1 << k)) ? __fmul_rn(ww, fr[k]) : __fmul_rn(ww, __fsub_rn(1.0f, fr[k]));
    w[c] = ww;
  }
}

// tcnn Frequency: out[d * 2n + 2k + {0, 1}] = sin(2^k pi x_d + {0, pi/2})
__device__ inline h16 freq_feature(int n_freq, const float* xr, int j) {
  int d = j / (2 * n_freq), k = (j / 2) % n_freq;
  float arg = __fmul_rn(__fmul_rn(xr[d], (float)(1 << k)), 3.14159265358979323846f);
  if (j & 1) arg = __fadd_rn(arg, 1.57079632679489661923f);
  return (h16)sinf(arg);
}

// Features of level l for the 64 samples of a wave (lane = sample, base_row = row of lane 0).
__device__ inline h16x2 level_features(const EncParams& p, int l, const float* xn, int lane, int base_row) {
  float w[8]; uint32_t cell[3];
  grid_weights(p.e.grid, l, xn, w, cell);
  const uint32_t* tab = p.table + p.e.grid.offset[l];
  // run-dedupe of the gathers: consecutive samples in the same cell read the same 8 entries; only the first sample
  // of a run (its head) loads them
  uint32_t q0 = __shfl_up(cell[0], 1), q1 = __shfl_up(cell[1], 1), q2 = __shfl_up(cell[2], 1);
  const bool head = (lane == 0) | (cell[0] != q0) | (cell[1] != q1) | (cell[2] != q2);
  const unsigned long long hm = __ballot(head);
  const int hl = 63 - __clzll(hm & ((2ull << lane) - 1ull));   // head lane of this lane's run
  // A gather costs one request per distinct 64-byte chunk per wave instruction (scripts/dev/probe_gather_pairs.hip),
  // and the two x-neighbour corners of a cell sit in one chunk 15 times out of 16 (x prime = 1, 16 fp16x2 entries per
  // chunk).  So the loads are issued in PAIR layout -- lanes 2j and 2j+1 fetch the x = 0 / x = 1 corner of sample
  // j (+32 in the second half) -- which halves the requests; the values then travel back to the owning lane.
  const uint32_t res = p.e.grid.res[l], size = p.e.grid.size[l];
  const bool dense = p.e.grid.dense[l] != 0;
  const uint32_t ty = dense ? res : PRIME_Y, tz = dense ? res * res : PRIME_Z;
  uint32_t r[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int s = (lane >> 1) + 32 * h;
    const uint32_t cx = __shfl(cell[0], s) + (uint32_t)(lane & 1), cy = __shfl(cell[1], s), cz = __shfl(cell[2], s);
    const bool act = ((hm >> s) & 1ull) && (base_row + s < p.rows);
    const uint32_t ay[2] = {cy * ty, cy * ty + ty}, az[2] = {cz * tz, cz * tz + tz};   // see grid_corners
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t ix;
      if (dense) { ix = cx + ay[i & 1] + az[i >> 1]; ix = ix >= size ? ix - size : ix; }
      else ix = (cx ^ ay[i & 1] ^ az[i >> 1]) & (size - 1u);
      r[h][i] = act ? tab[ix] : 0u;
    }
  }
  const int src = 2 * (hl & 31);
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {   // corner c = x | y << 1 | z << 2 ; accumulation order c = 0..7 as in the oracle
    const uint32_t t0 = __shfl(r[0][c >> 1], src + (c & 1)), t1 = __shfl(r[1][c >> 1], src + (c & 1));
    const uint32_t vv = (hl >> 5) ? t1 : t0;
    h16x2 hv = *(const h16x2*)&vv;
    a0 = __fadd_rn(a0, __fmul_rn(w[c], (float)hv[0]));
    a1 = __fadd_rn(a1, __fmul_rn(w[c], (float)hv[1]));
  }
  h16x2 o; o[0] = (h16)a0; o[1] = (h16)a1;
  return o;
}

__global__ __launch_bounds__(256) void k_encode_fwd(EncParams p, h16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* tile = (h16*)smem;  // [ENC_TILE][enc_pad]
  const int pad = p.e.enc_pad;
  const int fdim = 3 * 2 * p.e.n_freq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (p.rows + ENC_TILE - 1) / ENC_TILE;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int row = t * ENC_TILE + lane;
    bool valid = row < p.rows;
    float x[3] = {0, 0, 0}, xn[3];
    if (valid) row_position(p, row, x);
    if (p.e.use_grid) {
      normalize_pos(x, p.e.bound, true, xn);
      for (int l = p.level_lo + wave; l < p.level_hi; l += 4)
        *(h16x2*)&tile[lane * pad + fdim + 2 * l] = level_features(p, l, xn, lane, t * ENC_TILE);
    }
    // frequency part + ones padding: (row, j) pairs spread over the block
    const int extra0 = p.e.enc_dim;  // padding starts here
    const int nfj = fdim + (pad - extra0);
    for (int i = threadIdx.x; i < ENC_TILE * nfj; i += 256) {
      int r = i / nfj, j = i % nfj;
      if (j >= fdim) { tile[r * pad + extra0 + (j - fdim)] = (h16)1.0f; continue; }
      int rr = t * ENC_TILE + r;
      float xr[3] = {0, 0, 0};
      if (rr < p.rows) row_position(p, rr, xr);
      if (p.e.freq_normalized) { float q[3]; normalize_pos(xr, p.e.bound, false, q); xr[0] = q[0]; xr[1] = q[1]; xr[2] = q[2]; }
      tile[r * pad + j] = freq_feature(p.e.n_freq, xr, j);
    }
    __syncthreads();
    // coalesced write-out of the whole tile (rows are contiguous in memory)
    int rows_here = min(ENC_TILE, p.rows - t * ENC_TILE);
    int n16 = rows_here * pad / 8;
    uint4* dst = (uint4*)(out + (size_t)t * ENC_TILE * pad);
    const uint4* src = (const uint4*)tile;
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __syncthreads();
  }
}

// ---------------------------------------------------------------- level-phased forward
// scripts/dev/probe_encode_fwd_levels.py: with all 16 levels in flight the gathers miss the 4 MB L2 of every XCD (28.5 MB of
// tables) and the kernel runs at the rate the Infinity Cache delivers 64-byte lines: 21 us per level and 512 K samples.
// One or two levels at a time stay L2-resident: 6 us per level.  So for large row counts the levels are processed in PHASES:
// blocks are enumerated level-major (the dispatcher hands them out in order, so at any time the chip works on one or
// two neighbouring levels = 4 MB of table), every wave writes its 64 samples' features of a level as one
// coalesced 256-byte store into a per-level plane, and a second, streaming kernel assembles the row-major [rows, enc_pad]
// operand (frequency features, planes, ones) the MLP reads.
#define ENC_LG 1   // levels per phase (measured: 1 -> 135 us, 2 -> 140 us, 4 -> 202 us per 512 K samples; ALN_ENC_LG overrides)
__global__ __launch_bounds__(256) void k_encode_grid_phased(EncParams p, h16x2* __restrict__ planes, int nblk, int lg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x / nblk, b = blockIdx.x % nblk;
  const int base_row = (b * 4 + wave) * 64;
  if (base_row >= p.rows) return;
  const int row = base_row + lane;
  float x[3] = {0, 0, 0}, xn[3];
  if (row < p.rows) row_position(p, row, x);
  normalize_pos(x, p.e.bound, true, xn);
  for (int i = 0; i < lg; ++i) {
    const int l = g * lg + i;
    if (l < p.e.grid.n_levels) {
      const h16x2 o = level_features(p, l, xn, lane, base_row);
      if (row < p.rows) planes[(size_t)l * p.rows + row] = o;
    }
  }
}
__global__ __launch_bounds__(256) void k_encode_assemble(EncParams p, const h16x2* __restrict__ planes, h16* __restrict__ out) {
  // one thread per row builds it in an LDS tile (plane reads of a wave: 256 contiguous bytes per level); the block then
  // writes its 256 rows as one contiguous run of 16-byte pieces (row-per-lane 16-byte stores cost 2.5x the bytes in HBM
  // write traffic: WRITE_SIZE)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* tile = (h16*)smem;   // [256][pad + 8]: the +8 halves keep the row-per-lane 16-byte LDS stores off a 4-way conflict
  const int pad = p.e.enc_pad, fdim = 3 * 2 * p.e.n_freq, gdim = 2 * p.e.grid.n_levels, nch = pad / 8, tp = pad + 8;
  const int nblk = (p.rows + 255) / 256;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int row = b * 256 + threadIdx.x;
    if (row < p.rows) {
      float xr[3] = {0, 0, 0};
      if (fdim) {
        row_position(p, row, xr);
        if (p.e.freq_normalized) { float q[3]; normalize_pos(xr, p.e.bound, false, q); xr[0] = q[0]; xr[1] = q[1]; xr[2] = q[2]; }
      }
      for (int ch = 0; ch < nch; ++ch) {
        h16x8 v;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const int c = 8 * ch + j;   // fdim and gdim are even: a pair never straddles two parts
          if (c < fdim) { v[j] = freq_feature(p.e.n_freq, xr, c); v[j + 1] = freq_feature(p.e.n_freq, xr, c + 1); }
          else if (c < fdim + gdim) { const h16x2 f = planes[(size_t)((c - fdim) >> 1) * p.rows + row]; v[j] = f[0]; v[j + 1] = f[1]; }
          else { v[j] = (h16)1.0f; v[j + 1] = (h16)1.0f; }
        }
        *(h16x8*)(tile + threadIdx.x * tp + 8 * ch) = v;
      }
    }
    __syncthreads();
    const int rows_here = min(256, p.rows - b * 256);
    uint4* dst = (uint4*)(out + (size_t)b * 256 * pad);
    for (int i = threadIdx.x; i < rows_here * nch; i += 256) dst[i] = *(const uint4*)(tile + (i / nch) * tp + 8 * (i % nch));
    __syncthreads();
  }
}

static int fill_params(EncParams& p, const AlnEncDesc* e, const void* table, const float* rays_o, const float* rays_d,
                       const float* z, const float* xyz, int rows, int stride);
extern "C" int64_t aln_encode_fwd_ws_bytes(const AlnEncDesc* e, int32_t rows) {
  return (e && e->use_grid && rows > 0) ? (int64_t)e->grid.n_levels * rows * (int64_t)sizeof(h16x2) : 0;
}
extern "C" int aln_encode_fwd_phased(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d,
                                     const float* z, const float* xyz, int32_t rows, int32_t rays_stride, void* planes_ws,
                                     void* enc_out, void* stream) {
  EncParams p;
  if (int rc = fill_params(p, e, table_f16, rays_o, rays_d, z, xyz, rows, rays_stride)) return rc;
  ALN_REQUIRE(e->use_grid && table_f16 && planes_ws && enc_out, "encode_fwd_phased: needs a grid encoding, its table and the plane workspace");
  if (rows == 0) return 0;
  static const int lg = getenv("ALN_ENC_LG") ? atoi(getenv("ALN_ENC_LG")) : ENC_LG;
  const int nblk = (rows + 255) / 256, ngroups = ((int)e->grid.n_levels + lg - 1) / lg;
  hipLaunchKernelGGL(k_encode_grid_phased, dim3(nblk * ngroups), dim3(256), 0, (hipStream_t)stream, p, (h16x2*)planes_ws, nblk, lg);
  ALN_CHECK_LAUNCH("encode_grid_phased");
  hipLaunchKernelGGL(k_encode_assemble, dim3(nblk < 8192 ? nblk : 8192), dim3(256), 256 * (e->enc_pad + 8) * sizeof(h16), (hipStream_t)stream, p,
                     (const h16x2*)planes_ws, (h16*)enc_out);
  ALN_CHECK_LAUNCH("encode_assemble");
  return 0;
}

// Backward: scatter-add of w_c * dL/dfeat into the fp32 gradient table.
//
// Measured on MI355X (scripts/dev/probe_atomics*.hip): a global float atomic costs one request per distinct
// 32-byte sector per wave-instruction (~21 G requests/s chip-wide, independent of scope, XCD locality or table size);
// lanes hitting the SAME address serialize (14 G lane-ops/s), lanes sharing a sector coalesce (8 lanes/sector: 166 G
// lane-ops/s).  Hence:
//  1. run-dedupe: consecutive samples of a ray fall into the same cell at coarse levels (and, for the importance
//     samples, far into the fine levels); a wave-level segmented reduction sums them before anything is issued;
//  2. sector-aware issue: the 16 (corner, feature) adds of one cell are issued by 16 ADJACENT lanes ordered
//     [corner bit0 = x][feature], so the 2 features (8 B) and, when x is even, the x-pair (idx ^ 1: the x prime is 1)
//     share one 16-byte span -- 4..8 requests per cell instead of 16.
__global__ __launch_bounds__(256) void k_encode_bwd(EncParams p, const h16* __restrict__ d_enc, float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pad = p.e.enc_pad;
  const int fdim = 3 * 2 * p.e.n_freq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h16* tile = (h16*)smem;                                              // [ENC_TILE][pad]
  float* sval = (float*)(smem + ((ENC_TILE * pad * 2 + 15) & ~15)) + wave * (64 * 16);   // per wave [64 runs][16]
  uint32_t* sidx = (uint32_t*)((float*)(smem + ((ENC_TILE * pad * 2 + 15) & ~15)) + 4 * 64 * 16) + wave * (64 * 8);
  const int ntiles = (p.rows + ENC_TILE - 1) / ENC_TILE;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int rows_here = min(ENC_TILE, p.rows - t * ENC_TILE);
    int n16 = rows_here * pad / 8;
    const uint4* src = (const uint4*)(d_enc + (size_t)t * ENC_TILE * pad);
    uint4* dst = (uint4*)tile;
    for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    __syncthreads();
    const int row = t * ENC_TILE + lane;
    const bool valid = row < p.rows;
    float x[3] = {0, 0, 0}, xn[3];
    if (valid) row_position(p, row, x);
    normalize_pos(x, p.e.bound, true, xn);
    for (int l = p.level_lo + wave; l < p.level_hi; l += 4) {
      float g0 = 0.f, g1 = 0.f;
      if (valid) { h16x2 g = *(h16x2*)&tile[lane * pad + fdim + 2 * l]; g0 = (float)g[0]; g1 = (float)g[1]; }
      uint32_t idx[8]; float w[8];
      uint32_t cell[3];
      grid_corners(p.e.grid, l, xn, idx, w, cell);
      // a run = adjacent lanes in the same cell (same base corner => same 8 indices)
      // (shuffles are evaluated unconditionally: no short-circuit around cross-lane ops)
      uint32_t q0 = __shfl_up(cell[0], 1), q1 = __shfl_up(cell[1], 1), q2 = __shfl_up(cell[2], 1);
      bool head = (lane == 0) | (cell[0] != q0) | (cell[1] != q1) | (cell[2] != q2);
      unsigned long long hm = __ballot(head);
      int rid = __popcll(hm & ((2ull << lane) - 1ull)) - 1;
      float v[16];
#pragma unroll
      for (int c = 0; c < 8; ++c) { v[2 * c] = w[c] * g0; v[2 * c + 1] = w[c] * g1; }
      const int nruns = __popcll(hm);
      if (nruns < 64) {  // segmented sum towards the run head
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          int r2 = __shfl_down(rid, off);
          bool take = (lane + off < 64) && (r2 == rid);
#pragma unroll
          for (int k = 0; k < 16; ++k) { float o = __shfl_down(v[k], off); v[k] += take ? o : 0.f; }
        }
      }
      if (head) {
#pragma unroll
        for (int k = 0; k < 16; k += 4) *(float4*)&sval[rid * 16 + k] = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
#pragma unroll
        for (int c = 0; c < 8; c += 4) *(uint4*)&sidx[rid * 8 + c] = make_uint4(idx[c], idx[c + 1], idx[c + 2], idx[c + 3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float* gt = grad + 2 * (size_t)p.e.grid.offset[l];
      const int nops = nruns * 16;
      for (int op = lane; op < nops; op += 64) {
        int r = op >> 4, k = op & 15;
        float val = sval[r * 16 + k];
        if (val != 0.f) unsafeAtomicAdd(gt + 2 * (size_t)sidx[r * 8 + (k >> 1)] + (k & 1), val);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
}

static int fill_params(EncParams& p, const AlnEncDesc* e, const void* table, const float* rays_o, const float* rays_d,
                       const float* z, const float* xyz, int rows, int stride) {
  ALN_REQUIRE(e && rows >= 0, "encode: bad arguments");
  ALN_REQUIRE(xyz || (rays_o && rays_d && z && stride > 0), "encode: need xyz or rays+z");
  ALN_REQUIRE(e->enc_pad % 8 == 0 && e->enc_pad >= e->enc_dim, "encode: enc_pad must be a multiple of 8");
  ALN_REQUIRE(!e->use_grid || e->grid.n_features == 2, "encode: n_features_per_level must be 2");
  ALN_REQUIRE(e->enc_dim == 6 * e->n_freq + (e->use_grid ? 2 * e->grid.n_levels : 0), "encode: enc_dim mismatch");
  p.e = *e; p.table = (const uint32_t*)table; p.rays_o = rays_o; p.rays_d = rays_d; p.z = z; p.xyz = xyz;
  p.rows = rows; p.rays_stride = stride;
  p.level_lo = 0; p.level_hi = e->use_grid ? (int)e->grid.n_levels : 0;
  return 0;
}

extern "C" int aln_encode_fwd(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d,
                              const float* z, const float* xyz, int32_t rows, int32_t rays_stride, void* enc_out,
                              void* stream) {
  EncParams p;
  if (int rc = fill_params(p, e, table_f16, rays_o, rays_d, z, xyz, rows, rays_stride)) return rc;
  ALN_REQUIRE(!e->use_grid || table_f16, "encode_fwd: table is NULL");
  if (rows == 0) return 0;
  int ntiles = (rows + ENC_TILE - 1) / ENC_TILE;
  int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
  size_t lds = (size_t)ENC_TILE * e->enc_pad * sizeof(h16);
  hipLaunchKernelGGL(k_encode_fwd, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, (h16*)enc_out);
  ALN_CHECK_LAUNCH("encode_fwd");
  return 0;
}

// dev probe (scripts/dev/probe_encode_fwd_levels.py): forward for levels [level_lo, level_hi) only (other columns stay unwritten)
extern "C" int aln_dev_encode_fwd_levels(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d,
                                         const float* z, const float* xyz, int32_t rows, int32_t rays_stride, void* enc_out,
                                         int32_t level_lo, int32_t level_hi, void* stream) {
  EncParams p;
  if (int rc = fill_params(p, e, table_f16, rays_o, rays_d, z, xyz, rows, rays_stride)) return rc;
  if (rows == 0) return 0;
  p.level_lo = level_lo; p.level_hi = level_hi;
  int ntiles = (rows + ENC_TILE - 1) / ENC_TILE;
  int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
  size_t lds = (size_t)ENC_TILE * e->enc_pad * sizeof(h16);
  hipLaunchKernelGGL(k_encode_fwd, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, (h16*)enc_out);
  ALN_CHECK_LAUNCH("encode_fwd_levels");
  return 0;
}

extern "C" int aln_encode_bwd_levels(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                                     const float* xyz, int32_t rows, int32_t rays_stride, const void* d_enc,
                                     float* grad_table, int32_t level_lo, int32_t level_hi, void* stream) {
  EncParams p;
  if (int rc = fill_params(p, e, nullptr, rays_o, rays_d, z, xyz, rows, rays_stride)) return rc;
  if (rows == 0 || !e->use_grid) return 0;
  ALN_REQUIRE(0 <= level_lo && level_lo <= level_hi && level_hi <= (int)e->grid.n_levels, "encode_bwd: level range [%d, %d)", level_lo,
              level_hi);
  if (level_lo == level_hi) return 0;
  p.level_lo = level_lo; p.level_hi = level_hi;
  int ntiles = (rows + ENC_TILE - 1) / ENC_TILE;
  int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
  size_t lds = (((size_t)ENC_TILE * e->enc_pad * sizeof(h16) + 15) & ~(size_t)15) + 4 * 64 * (16 * sizeof(float) + 8 * sizeof(uint32_t));
  hipLaunchKernelGGL(k_encode_bwd, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, (const h16*)d_enc, grad_table);
  ALN_CHECK_LAUNCH("encode_bwd");
  return 0;
}

extern "C" int aln_encode_bwd(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                              const float* xyz, int32_t rows, int32_t rays_stride, const void* d_enc,
                              float* grad_table, void* stream) {
  ALN_REQUIRE(e, "encode_bwd: NULL descriptor");
  return aln_encode_bwd_levels(e, rays_o, rays_d, z, xyz, rows, rays_stride, d_enc, grad_table, 0, (int32_t)e->grid.n_levels, stream);
}

extern "C" int aln_grid_desc_init(AlnGridDesc* g) {
  ALN_REQUIRE(g && g->n_levels > 0 && g->n_levels <= ALN_MAX_LEVELS, "grid_desc: n_levels out of range");
  uint32_t offset = 0;
  for (int l = 0; l < g->n_levels; ++l) {
    // tcnn: scale = exp2(l * log2(pls)) * base - 1 (fp32); res = ceil(scale) + 1
    float scale = exp2f((float)l * log2f(g->per_level_scale)) * (float)g->base_resolution - 1.0f;
    uint32_t res = (uint32_t)ceilf(scale) + 1u;
    uint64_t dense = (uint64_t)res * res * res;
    uint64_t size = dense > 0x7FFFFFFFull ? 0x7FFFFFFFull : dense;
    size = (size + 7) / 8 * 8;
    uint64_t cap = 1ull << g->log2_hashmap_size;
    if (size > cap) size = cap;
    g->scale[l] = scale; g->res[l] = res; g->size[l] = (uint32_t)size; g->offset[l] = offset;
    g->dense[l] = dense <= size ? 1u : 0u;
    offset += (uint32_t)size;
  }
  g->n_entries = offset;
  return 0;
}
