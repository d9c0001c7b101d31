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
  // cell mode (density-grid refresh of the marching path, march.hip): row r is one jittered point of occupancy-grid cell
  // cell0 + r -- the positions of aln_grid_points, generated in place instead of being read from a [G^3, 3] buffer
  int cell_G, cell0; uint32_t cell_seed, cell_step; const uint32_t* cell_step_dev;
};

__device__ inline void row_position(const EncParams& p, int row, float* x) {
  if (p.cell_G) {
    const uint32_t key = aln_rand_key(p.cell_seed, ALN_STREAM_PERTURB, p.cell_step + (p.cell_step_dev ? *p.cell_step_dev : 0u));
    const uint32_t g = (uint32_t)p.cell_G, c = (uint32_t)(p.cell0 + row);
    const uint32_t r = c / g;
    const float cf[3] = {(float)(c % g), (float)(r % g), (float)(r / g)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float u = aln_rand_uniform(key, 3u * c + (uint32_t)k);
      x[k] = __fsub_rn(__fmul_rn(__fdiv_rn(__fadd_rn(cf[k], u), (float)p.cell_G), __fmul_rn(2.0f, p.e.bound)), p.e.bound);
    }
    return;
  }
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

// level-local corner indices + trilinear weights (oracle: grid_corner_indices).  KIND: -1 = dense / hashed decided per call (a
// wave-uniform branch per corner), 0 = hashed, 1 = dense (the caller branches ONCE per level: k_encode_bwd_bin)
template <int KIND = -1>
__device__ inline void grid_corners(const AlnGridDesc& g, int l, const float* xn, uint32_t* idx, float* w, uint32_t* base = nullptr) {
  float scale = g.scale[l];
  uint32_t res = g.res[l], size = g.size[l];
  const bool dense = KIND < 0 ? g.dense[l] != 0 : KIND == 1;
  uint32_t gi[3]; float fr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pos = g.pos_fma ? __fmaf_rn(xn[k], scale, 0.5f) : __fadd_rn(__fmul_rn(xn[k], scale), 0.5f);
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
    float pos = g.pos_fma ? __fmaf_rn(xn[k], scale, 0.5f) : __fadd_rn(__fmul_rn(xn[k], scale), 0.5f);
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
#ifdef STUB_GATHER_NOLOAD     // dev stub: no table loads
      r[h][i] = act ? ix * 2654435761u : 0u;
#else
      r[h][i] = act ? tab[ix] : 0u;   // (round 6: `nt` on these loads 145 -> 365 us -- they leave the L2; sc0 / sc1 / both: 146-150 us)
#endif
    }
  }
  const int src = 2 * (hl & 31);
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {   // corner c = x | y << 1 | z << 2 ; accumulation order c = 0..7 as in the oracle
#ifdef STUB_GATHER_NORETURN   // dev stub: no return shuffles (wrong values)
    const uint32_t t0 = r[0][c >> 1], t1 = r[1][c >> 1];
#else
    const uint32_t t0 = __shfl(r[0][c >> 1], src + (c & 1)), t1 = __shfl(r[1][c >> 1], src + (c & 1));
#endif
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
#define ENC_LG 1   // levels per phase (measured: 1 -> 135 us, 2 -> 140 us, 4 -> 202 us per 512 K samples)
#ifndef ENC_LG_TILED
#define ENC_LG_TILED 1   // tiled output (2: one 8-byte store per lane for the two levels of a phase -- measured 299 us against 288 per step)
#endif
// `tiled` != NULL (round 5): no planes, no assembly pass -- the wave writes its level's features straight into the MLP's input buffer in
// the TILED layout the 128-wide kernels' loaders fetch (AlnMlpDesc.x_tiled: 16-byte piece q of row r at 32 pad (r / 32) + 256 q + 8 (r % 32)
// halves).  A 4-byte store per lane at a 16-byte stride fills a quarter of each line; the other three levels of the piece come from
// blocks with the same row range a few microseconds later, and those run on the SAME XCD (block ids differ by multiples of nblk, a
// multiple of 8), so the pieces merge in its L2 before they leave for HBM.  The blocks of the first phase also write the frequency
// features and the ones padding of their rows.
// `pitch` != 0 (round 6): PAIR-PLANE output for the density head itself (AlnMlpDesc.x_tiled = 2) -- level l goes to plane fdim / 2 + l at
// `pitch` words per plane, and the phases also write the frequency pairs (planes 0 .. fdim / 2 - 1) and the ones behind the last level,
// one pair per phase like the tiled form: every plane is written with whole-wave 256-byte stores, nothing is assembled afterwards.
// (round 6: several consecutive 256-row pieces of a level per block -- fewer, longer blocks -- measured 149 / 167 / 245 us for 2 / 4 / 8
//  pieces against 146 us: the 32 768 independent short blocks are what hides the gather latency)
__global__ __launch_bounds__(256) void k_encode_grid_phased(EncParams p, h16x2* __restrict__ planes, int nblk, int lg, h16* __restrict__ tiled, long pitch) {
  const int g = blockIdx.x / nblk, b = blockIdx.x % nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base_row = (b * 4 + wave) * 64;
  if (base_row >= p.rows) return;
  const int row = base_row + lane;
  float x[3] = {0, 0, 0}, xn[3];
  if (row < p.rows) row_position(p, row, x);
  normalize_pos(x, p.e.bound, true, xn);
  const int pad = p.e.enc_pad, fdim = 3 * 2 * p.e.n_freq;
  h16* const trow = tiled ? tiled + (size_t)(row >> 5) * (32 * pad) + 8 * (row & 31) : nullptr;   // piece q of this row: trow + 256 q
  if (tiled && lg == 2 && g * 2 + 1 < p.e.grid.n_levels && ((fdim + 4 * g) & 3) == 0) {   // the phase's two levels: ONE 8-byte store
    const h16x2 o0 = level_features(p, 2 * g, xn, lane, base_row), o1 = level_features(p, 2 * g + 1, xn, lane, base_row);
    if (row < p.rows) {
      const int j = fdim + 4 * g;
      h16x4 o; o[0] = o0[0]; o[1] = o0[1]; o[2] = o1[0]; o[3] = o1[1];
      *(h16x4*)(trow + 256 * (j >> 3) + (j & 7)) = o;
    }
  } else
  for (int i = 0; i < lg; ++i) {
    const int l = g * lg + i;
    if (l < p.e.grid.n_levels) {
      const h16x2 o = level_features(p, l, xn, lane, base_row);
      if (row < p.rows) {
        if (tiled) { const int j = fdim + 2 * l; *(h16x2*)(trow + 256 * (j >> 3) + (j & 7)) = o; }
        else if (pitch) planes[(size_t)(fdim / 2 + l) * pitch + row] = o;
        else planes[(size_t)l * p.rows + row] = o;
      }
    }
  }
  if ((tiled || pitch) && row < p.rows) {
    // frequency features (k_encode_assemble's arithmetic) and the ones behind the last level: one PAIR of columns per phase -- phase g
    // takes pair g of the fdim / 2 + (pad - enc_dim) / 2 extra pairs (all twelve sines in the first phase made its blocks three times
    // as long as the others); whatever does not fit the phases falls to the last one
    const int npf = fdim / 2, npo = (pad - p.e.enc_dim) / 2, ngroups = (p.e.grid.n_levels + lg - 1) / lg;
    for (int q = g; q < npf + npo; q += ngroups) {
      if (q < npf) {
        float xr[3] = {x[0], x[1], x[2]};
        if (p.e.freq_normalized) { float qn[3]; normalize_pos(xr, p.e.bound, false, qn); xr[0] = qn[0]; xr[1] = qn[1]; xr[2] = qn[2]; }
        const int j = 2 * q;
#ifdef STUB_FREQ   // dev stub: no sines
        h16x2 f; f[0] = (h16)xr[0]; f[1] = (h16)xr[1];
#else
        h16x2 f; f[0] = freq_feature(p.e.n_freq, xr, j); f[1] = freq_feature(p.e.n_freq, xr, j + 1);
#endif
        if (tiled) *(h16x2*)(trow + 256 * (j >> 3) + (j & 7)) = f; else planes[(size_t)q * pitch + row] = f;
      } else {
        const int j = p.e.enc_dim + 2 * (q - npf);
        h16x2 one; one[0] = (h16)1.0f; one[1] = (h16)1.0f;
        if (tiled) *(h16x2*)(trow + 256 * (j >> 3) + (j & 7)) = one; else planes[(size_t)(j >> 1) * pitch + row] = one;
      }
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
  ALN_REQUIRE(e->use_grid && table_f16 && enc_out, "encode_fwd_phased: needs a grid encoding and its table");
  ALN_REQUIRE(planes_ws || ((e->enc_pad == 32 || e->enc_pad == 48) && e->enc_dim % 2 == 0 && ((uintptr_t)enc_out & 15) == 0),
              "encode_fwd_phased: the tiled output (planes_ws = NULL) needs enc_pad 32 or 48");
  if (rows == 0) return 0;
  const int lg = planes_ws ? ENC_LG : ENC_LG_TILED;
  const int nblk = (rows + 255) / 256, ngroups = ((int)e->grid.n_levels + lg - 1) / lg;
  if (!planes_ws) {   // tiled: straight into the MLP's input buffer (nblk rounded up to a multiple of 8 keeps a row range on one XCD)
    const int nblk8 = (nblk + 7) / 8 * 8;
    hipLaunchKernelGGL(k_encode_grid_phased, dim3(nblk8 * ngroups), dim3(256), 0, (hipStream_t)stream, p, (h16x2*)nullptr, nblk8, lg, (h16*)enc_out, 0L);
    ALN_CHECK_LAUNCH("encode_grid_phased");
    return 0;
  }
  hipLaunchKernelGGL(k_encode_grid_phased, dim3(nblk * ngroups), dim3(256), 0, (hipStream_t)stream, p, (h16x2*)planes_ws, nblk, lg, (h16*)nullptr, 0L);
  ALN_CHECK_LAUNCH("encode_grid_phased");
  hipLaunchKernelGGL(k_encode_assemble, dim3(nblk < 8192 ? nblk : 8192), dim3(256), 256 * (e->enc_pad + 8) * sizeof(h16), (hipStream_t)stream, p,
                     (const h16x2*)planes_ws, (h16*)enc_out);
  ALN_CHECK_LAUNCH("encode_assemble");
  return 0;
}

extern "C" int aln_encode_fwd_planes(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d, const float* z,
                                     const float* xyz, int32_t rows, int32_t rays_stride, void* planes_out, int64_t plane_pitch, void* stream) {
  EncParams p;
  if (int rc = fill_params(p, e, table_f16, rays_o, rays_d, z, xyz, rows, rays_stride)) return rc;
  ALN_REQUIRE(e->use_grid && table_f16 && planes_out, "encode_fwd_planes: needs a grid encoding, its table and the plane buffer");
  ALN_REQUIRE(plane_pitch >= rows && plane_pitch % 4 == 0 && ((uintptr_t)planes_out & 3) == 0 && e->enc_dim % 2 == 0 && (6 * e->n_freq) % 2 == 0,
              "encode_fwd_planes: plane_pitch must be a multiple of 4 words and at least the row count");
  if (rows == 0) return 0;
  const int nblk = (rows + 255) / 256, ngroups = (int)e->grid.n_levels;
  hipLaunchKernelGGL(k_encode_grid_phased, dim3(nblk * ngroups), dim3(256), 0, (hipStream_t)stream, p, (h16x2*)planes_out, nblk, 1, (h16*)nullptr, (long)plane_pitch);
  ALN_CHECK_LAUNCH("encode_grid_phased");
  return 0;
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
  p.cell_G = 0; p.cell0 = 0; p.cell_seed = p.cell_step = 0; p.cell_step_dev = nullptr;
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

// encoding of one jittered point per occupancy-grid cell (cells [cell0, cell0 + rows) of a G^3 grid): the input of the density
// head for NeRFRenderer.update_extra_state (march.hip); same positions as aln_grid_points, never stored
extern "C" int aln_encode_fwd_cells(const AlnEncDesc* e, const void* table_f16, int32_t G, uint32_t seed, uint32_t step,
                                    const uint32_t* step_dev, int32_t cell0, int32_t rows, void* planes_ws, void* enc_out,
                                    void* stream) {
  EncParams p;
  static const float dummy = 0.f;   // fill_params wants a position source; cell mode overrides it
  if (int rc = fill_params(p, e, table_f16, nullptr, nullptr, nullptr, &dummy, rows, 1)) return rc;
  ALN_REQUIRE(G > 0 && G <= 1024 && cell0 >= 0 && (int64_t)cell0 + rows <= (int64_t)G * G * G && enc_out, "encode_fwd_cells: bad cell range");
  ALN_REQUIRE(!e->use_grid || table_f16, "encode_fwd_cells: table is NULL");
  p.xyz = nullptr; p.cell_G = G; p.cell0 = cell0; p.cell_seed = seed; p.cell_step = step; p.cell_step_dev = step_dev;
  if (rows == 0) return 0;
  if (e->use_grid && planes_ws) {
    const int lg = ENC_LG;
    const int nblk = (rows + 255) / 256, ngroups = ((int)e->grid.n_levels + lg - 1) / lg;
    hipLaunchKernelGGL(k_encode_grid_phased, dim3(nblk * ngroups), dim3(256), 0, (hipStream_t)stream, p, (h16x2*)planes_ws, nblk, lg, (h16*)nullptr, 0L);
    ALN_CHECK_LAUNCH("encode_grid_phased");
    hipLaunchKernelGGL(k_encode_assemble, dim3(nblk < 8192 ? nblk : 8192), dim3(256), 256 * (e->enc_pad + 8) * sizeof(h16), (hipStream_t)stream, p,
                       (const h16x2*)planes_ws, (h16*)enc_out);
    ALN_CHECK_LAUNCH("encode_assemble");
    return 0;
  }
  int ntiles = (rows + ENC_TILE - 1) / ENC_TILE;
  int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
  size_t lds = (size_t)ENC_TILE * e->enc_pad * sizeof(h16);
  hipLaunchKernelGGL(k_encode_fwd, dim3(grid), dim3(256), lds, (hipStream_t)stream, p, (h16*)enc_out);
  ALN_CHECK_LAUNCH("encode_fwd");
  return 0;
}

// ---------------------------------------------------------------- binned backward (no global atomics)
// A scatter through global fp32 atomics is bound by the L2 atomic units (~21 G 64-byte requests/s, profiles/r02_probe_atomics.txt):
// every (sample, level) costs ~4 requests although each 64-byte chunk of a 4 MB level table is hit ~64 times per step (round 1's
// kernel; it also made every run order-dependent).  The binned backward turns the scatter into two streaming passes:
//   phase 1 (k_encode_bwd_bin): a block owns a tile of 512 consecutive sample rows.  Per level it computes the run-deduped
//     (index, w * dL/dfeat) records, counting-sorts them in LDS by table SLICE (1/64 of the level: 8192 entries for a hashed
//     level, 64 for the 16^3 level) and writes the sorted records as ONE contiguous run into the tile's fixed chunk of the record
//     pool, plus one (start, count, shift) descriptor per slice.  Round 6: a record is a PAIR -- the two x-neighbour corners of a cell,
//     which practically always fall into one slice -- in 12 bytes: slot0 | slot1 << 13, fp16x2 value 0, fp16x2 value 1 (round 5: two
//     8-byte records).  The copy-out of the pool through LDS is 40 % of phase 1 and proportional to its bytes (stub: -30 us).
//   phase 2 (k_encode_bwd_accum): a block owns one (level, slice); it streams that slice's runs of every tile, accumulates
//     them with 64-bit integer LDS atomics (exact, order-independent) and adds the slice to the gradient table with plain
//     coalesced stores.  Every level has the same 64-way split, so no entry is ever shared by two blocks: the whole gradient
//     table is bit-reproducible run to run (round 2 split the tiles of the two coarsest levels over blocks that met in fp32
//     global atomics).
// HBM traffic: 12 B written + 12 B read per pair record (<= 4 x 16 per sample) instead of 4 atomic requests per (sample, level).
#ifndef BIN_TILE
#define BIN_TILE 512            // sample rows per phase-1 block (= threads: lane = sample, so runs along a ray dedupe in-wave); -DBIN_TILE=1024
                                // (1 KB runs for phase 2) measured: round 3 dense pair 771 -> 806 us, round 5 620 -> 607 us (step -0.3 %): no gain
                                // worth the format change it needs -- a full 1024-row tile has 8192 records, one more than DESC_START holds
#endif
static_assert(BIN_TILE * 8 <= 0x1FFF, "a tile's worst-case pair-record count must fit the 13-bit start field of the descriptors");
// descriptor word: start (13 bits) | count << 13 (14 bits) | (shift + BIN_SHIFT_BIAS) << 27
#define DESC_START(q) ((q) & 0x1FFFu)
#define DESC_COUNT(q) (((q) >> 13) & 0x3FFFu)
#define BIN_SHIFT_BIAS 8
#define BIN_MIN_SHIFT (-8)      // a run sums <= 64 products of magnitude <= 65504: below 2^23, so scaling down by 2^8 always fits fp16
#define DESC_SHIFT(q) ((int)((q) >> 27) - BIN_SHIFT_BIAS)
#define BIN_SLICE_LOG2 13       // largest slice: 8192 table entries (x 2 features x 8 B = 128 KB of LDS accumulators in phase 2)
#define BIN_SLICE (1 << BIN_SLICE_LOG2)
#define BIN_MAX_SLICES 64       // slices per level (2^19 entries / 8192)
#define BIN_CHUNK (BIN_TILE * 4)   // PAIR records per (tile, level) chunk of the pool
#define BIN_REC_WORDS 3             // 4-byte words per pair record: [slot0 | slot1 << 13] [fp16x2 value 0] [fp16x2 value 1] -- an ODD record of a chunk
                                   // keeps its slot word FIRST, an even one LAST, so that the 8 value bytes are 8-byte aligned in both (LDS stores)
// A pair whose two corners fall into DIFFERENT slices (x + 1 carries past the slice width: x = ...1111111111111 -- ~2^-13 of the pairs of a
// fine hashed level on continuous positions, but EVERY sample clamped onto the +x face of the box at the levels of 8192 cells and more:
// 16 % of the pairs of levels 9 .. 15 on the bench scene) becomes TWO records, each in the run of its own slice with an empty second half
// (value word 0: phase 2 skips it).  A tile's level can therefore hold up to 2 x BIN_CHUNK records: the pool's chunks have room for the
// worst case, the LDS tile for BIN_CHUNK -- records ranked beyond that (the highest slices of a tile full of clamped samples) go straight
// to their place in the pool with 12-byte stores instead of through the sorted LDS tile.
#define BIN_CHUNK_WORDS (2 * BIN_CHUNK * BIN_REC_WORDS)              // pitch of a (tile, level) chunk of the pool: 48 KB
#define BIN_MAX_SHIFT 11        // largest per-tile up-scaling of the fp16 record values (BIN_MIN_SHIFT: largest down-scaling)
#ifndef BIN_DEDUPE_LEVELS
#define BIN_DEDUPE_LEVELS 8
#endif                          // levels below this run the in-wave run-dedupe (finer: consecutive samples practically never share a cell)

// entries per slice of a level = 2^slice_log2: the level is cut into (at most) BIN_MAX_SLICES slices of at most BIN_SLICE entries.
// (Fewer, larger slices for the coarse dense levels -- longer runs for phase 2 -- measured worse: a phase-2 block sustains ~1.4 G
//  records/s whatever the run length, so the 5 blocks of a 1024-entry split of the 17^3 level became the long pole: 129 vs 80 us.)
static inline int bin_slice_log2(uint32_t size) {
  int lg = 0;
  while ((1ull << lg) < size) ++lg;
  lg -= 6;
  return lg < 0 ? 0 : (lg > BIN_SLICE_LOG2 ? BIN_SLICE_LOG2 : lg);
}

struct BinParams {
  EncParams p;
  const h16* d_enc;
  int32_t* found_inf; // raised by phase 1 when a run is non-finite (before phase 2 can step anything)
  uint32_t* pool;     // [n_levels][ntiles] chunks of BIN_CHUNK_WORDS words: up to 2 x BIN_CHUNK pair records of BIN_REC_WORDS words
  uint32_t* desc;     // [n_levels][BIN_MAX_SLICES][ntiles]  DESC_START | DESC_COUNT | DESC_SHIFT
  int ntiles;
  int rows1, stride2; // rows [0, rows1) use p.rays_stride samples per ray, the rest stride2 (coarse + fine pass in one launch)
  const uint16_t* perm;   // optional [rays][stride1 + stride2]: sample ids of a ray in depth order (sampling.hip) -- the tile then walks the
                          // samples in that order, so the coarse and the fine samples of one cell form ONE run of the in-wave dedupe
  uint32_t slice_log2_w[ALN_MAX_LEVELS / 4];   // slice_log2 of level l = byte l & 3 of word l >> 2
};

__device__ inline void bin_row_position(const BinParams& b, int row, float* x) {
  const EncParams& p = b.p;
  if (p.xyz) { x[0] = p.xyz[3 * (size_t)row]; x[1] = p.xyz[3 * (size_t)row + 1]; x[2] = p.xyz[3 * (size_t)row + 2]; return; }
  const int ray = row < b.rows1 ? row / p.rays_stride : (row - b.rows1) / b.stride2;
  aln_sample_xyz(p.rays_o + 3 * (size_t)ray, p.rays_d + 3 * (size_t)ray, p.z[row], p.e.bound, x);
}

// (round 5's value / timestamp taps for scripts/dev/stress_scatter.py --map-lib / --times -- -DBIN_DEBUG=5 / 6 builds -- are in the history of this file:
//  commit 74cae74)
__device__ inline uint32_t bin_pack_h2(float a, float b) {
  h16x2 h; h[0] = (h16)a; h[1] = (h16)b;
  return *(const uint32_t*)&h;
}
#ifdef BIN_TIMING   // dev builds only (scripts/dev/bench_scatter.py --timing): shader-clock ticks per section of phase 1, block BIN_TIMING_BLOCK, waves 0 and 7
__device__ long long g_bin_t[2][12];
extern "C" int aln_debug_read_bin_timing(long long* host_out, int reset) {
  if (reset) { long long z[24] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bin_t), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bin_t), sizeof(long long) * 24);
}
#define BT_DECL long long bt_acc[12] = {0}; long long bt_last = clock64();
#define BT(i) { long long bt_now = clock64(); bt_acc[i] += bt_now - bt_last; bt_last = bt_now; }
#define BT_FLUSH if (blockIdx.x == 300 && (tid == 0 || tid == 448)) { for (int i = 0; i < 12; ++i) g_bin_t[tid != 0][i] += bt_acc[i]; }
#else
#define BT_DECL
#define BT(i)
#define BT_FLUSH
#endif
// inclusive prefix sum over the 64 lanes of a wave through the DPP network: shifts inside the rows of 16, then the row broadcasts
__device__ inline uint32_t wave_incl_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
  return x;
}
__global__ __launch_bounds__(BIN_TILE) __attribute__((amdgpu_waves_per_eu(BIN_TILE <= 768 ? 6 : 4, BIN_TILE <= 768 ? 6 : 4))) void k_encode_bwd_bin(BinParams b) {   // (three blocks per CU: 80 registers)
  __shared__ __attribute__((aligned(16))) uint32_t sorted[BIN_CHUNK * BIN_REC_WORDS];   // 24 KB
  __shared__ uint32_t gws[8][BIN_TILE];                      // 16 KB: the rows' gradient words of eight levels (column = thread)
  __shared__ uint32_t cnt[2][BIN_MAX_SLICES], base_w[BIN_TILE / 64][BIN_MAX_SLICES], vmax_s[2];
  const EncParams& p = b.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = blockIdx.x;
  const int row0 = tile * BIN_TILE;
  const int pad = p.e.enc_pad, fdim = 3 * 2 * p.e.n_freq;
  // The row's gradient words (fp16x2 per level) are fetched eight levels at a time -- 32 contiguous bytes of the row -- and parked
  // in a thread-private LDS column.  (Round 2 read one word per level straight from d_enc "because the lines stay in L1 / L2":
  // the PMC pass of round 3 says they do not -- FETCH_SIZE 679 MB per launch for 100 MB of d_enc, the 48 KB tiles of the ~100
  // resident blocks of an XCD thrash its 4 MB L2 between two levels.  Staging all 16 levels at once costs 32 KB and a block per CU.)
  if (tid < 2 * BIN_MAX_SLICES) cnt[0][tid] = 0;
  if (tid < 2) vmax_s[tid] = 0u;
  const bool valid = row0 + tid < p.rows;
  int row = row0 + tid;
  if (b.perm && valid) {   // position row0 + tid of the depth-ordered walk -> sample row (pass-major layout)
    const int S1 = p.rays_stride, S = S1 + b.stride2, ray = row / S;
    const int id = b.perm[row];
    row = id < S1 ? ray * S1 + id : b.rows1 + ray * b.stride2 + (id - S1);
  }
  float x[3] = {0, 0, 0}, xn[3];
  if (valid) bin_row_position(b, row, x);
  normalize_pos(x, p.e.bound, true, xn);
  const uint32_t* const grow = (const uint32_t*)(b.d_enc + (size_t)(valid ? row : 0) * pad + fdim);
  const bool al8 = (fdim * 2) % 8 == 0 && (pad * 2) % 8 == 0;   // 8-byte loads when the grid part starts 8-byte aligned in every row
  // every load is requested before the first is parked (hipcc would otherwise wait for each one in turn), and -- round 6 -- the words of
  // levels l .. l + 7 are requested one level EARLY, after the sorted stores of level l - 1, so that their trip to HBM (8 000 ticks
  // per eight levels at the top of the level: 7 % of the kernel in the section clocks of a -DBIN_TIMING build) hides behind the stores,
  // the barrier and the copy-out; the eight registers are live where the sixteen products are not
  uint2 w2[4];
  // (unconditional loads from clamped addresses, masked where they are parked, a level later: with a load per branch hipcc waited for each
  //  in turn -- vmcnt(0), eight HBM round trips in a row and every store of the copy-out drained in front of the first)
  auto fetch_gw = [&](int l) __attribute__((always_inline)) {
    const int nw = min(8, p.level_hi - l);
    if (al8 && (l & 1) == 0 && (nw & 1) == 0) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) w2[j >> 1] = *(const uint2*)(grow + l + min(j, nw - 2));
    } else {
#pragma unroll
      for (int j = 0; j < 8; j += 2) { w2[j >> 1].x = grow[l + min(j, nw - 1)]; w2[j >> 1].y = grow[l + min(j + 1, nw - 1)]; }
    }
  };
  fetch_gw(p.level_lo);
  __syncthreads();
  BT_DECL
  // The sorted tile of level l leaves for the record pool in 3 pieces of 16 bytes per thread (24 KB), SPREAD OVER THE
  // ARITHMETIC OF LEVEL l + 1 (round 6).  As one burst behind the last barrier of its level -- every block of the chip at about the
  // same time -- the then 743 MB of 8-byte records cost the kernel 130 us of its 315 (stub: no copy-out, 185 us): the memory pipe idles while
  // the waves compute and the waves stall on full store queues while it drains.  `sorted` is not written again before the first barrier
  // of level l + 1, which no wave reaches before its last piece has been read.
  uint32_t cp_n = 0u;          // 16-byte pieces of the level whose copy-out is pending
  uint4* cp_dst = nullptr;
  auto copy_piece = [&](int i) __attribute__((always_inline)) {
    const uint32_t j = (uint32_t)tid + (uint32_t)i * BIN_TILE;
#if defined(STUB_COPY)          // dev stubs: no copy-out at all / LDS reads without stores / stores without LDS reads
    (void)j;
#elif defined(STUB_COPY_NOSTORE)
    if (j < cp_n) { const uint4 q = ((const uint4*)sorted)[j]; if (q.x == 0xdeadbeefu && q.y == 0x12345678u) cp_dst[j] = q; }
#elif defined(STUB_COPY_NOLDS)
    if (j < cp_n) cp_dst[j] = make_uint4(j, tid, i, 0u);
#else
    if (j < cp_n) cp_dst[j] = ((const uint4*)sorted)[j];
#endif
  };
  for (int l = p.level_lo; l < p.level_hi; ++l) {
    BT(0)
    const int par = l & 1;
    // (slice_log2 is read as a WORD of the kernel arguments and shifted: indexed as bytes hipcc fetched it with a vector load, and the
    //  s_waitcnt vmcnt(0) in front of its first use made every level wait for all of the wave's stores in flight -- the copy-out)
    const uint32_t sl = (b.slice_log2_w[l >> 2] >> (8 * (l & 3))) & 0xFFu, slot_mask = (1u << sl) - 1u;
    const int k8 = (l - p.level_lo) & 7;
    if (k8 == 0) {   // words of levels l .. l + 7 (requested a level ago: fetch_gw) -> this thread's column (nobody else touches it: no barrier)
      const int nw = min(8, p.level_hi - l);
#pragma unroll
      for (int j = 0; j < 8; j += 2) { gws[j][tid] = (valid && j < nw) ? w2[j >> 1].x : 0u; gws[j + 1][tid] = (valid && j + 1 < nw) ? w2[j >> 1].y : 0u; }
    }
    BT(1)
    copy_piece(0);
    const uint32_t gw = gws[k8][tid];
    const h16x2 g = *(const h16x2*)&gw;
    const float g0 = (float)g[0], g1 = (float)g[1];
    uint32_t idx[8]; float w[8]; uint32_t cell[3];
#ifdef STUB_COMPUTE   // dev stub: no corner arithmetic (cheap fake indices / weights)
#pragma unroll
    for (int c = 0; c < 8; ++c) { idx[c] = ((uint32_t)tid * 2654435761u + (uint32_t)(c >> 1) * 40503u + (uint32_t)l * 977u + (c & 1)) & (p.e.grid.size[l] - 1u); w[c] = 0.125f; }
    cell[0] = tid; cell[1] = l; cell[2] = 0;
#else
    if (p.e.grid.dense[l]) grid_corners<1>(p.e.grid, l, xn, idx, w, cell);   // one wave-uniform branch per level, not one per corner
    else grid_corners<0>(p.e.grid, l, xn, idx, w, cell);
#endif
    // run-dedupe along the ray: adjacent lanes in the same cell are summed into the run head
    // (levels from BIN_DEDUPE_LEVELS on skip it: at 4096 cells per axis and beyond consecutive samples practically never share
    //  a cell -- 8.1-8.3 of 8.4 M records survive -- so the compare / ballot / ladder step costs more than the records it saves)
    copy_piece(1);
    const bool dd = l < BIN_DEDUPE_LEVELS;
    uint32_t q0 = cell[0], q1 = cell[1], q2 = cell[2];
    if (dd) { q0 = __shfl_up(cell[0], 1); q1 = __shfl_up(cell[1], 1); q2 = __shfl_up(cell[2], 1); }
    const bool head = !dd | (lane == 0) | (cell[0] != q0) | (cell[1] != q1) | (cell[2] != q2);
    const unsigned long long hm = dd ? __ballot(head) : ~0ull;
    const int rid = __popcll(hm & ((2ull << lane) - 1ull)) - 1;
    float v[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[2 * c] = w[c] * g0; v[2 * c + 1] = w[c] * g1; }
    copy_piece(2);
    if (__popcll(hm) < 64) {
      // runs are contiguous: once no lane finds a run mate at distance `off`, none exists further away (most fine levels
      // leave after one step; the full ladder is 6 x 16 shuffles)
      for (int off = 1; off < 64; off <<= 1) {
        const int r2 = __shfl_down(rid, off);
        const bool take = (lane + off < 64) && (r2 == rid);
        if (!__any(take)) break;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float o = __shfl_down(v[k], off); v[k] += take ? o : 0.f; }
      }
    }
    // records of this lane (run heads only).  Values travel as fp16x2 scaled by 2^e, e = per (tile, level) exponent chosen so
    // that the largest |value| of the tile lands in [2^14, 2^15) (BIN_MIN_SHIFT <= e <= BIN_MAX_SHIFT): tiny products w * g keep
    // their bits instead of flushing at the fp16 denormal step, and a run sum beyond the fp16 range (64 samples of one cell at
    // the largest loss scales) is scaled down instead of overflowing -- a record is non-finite only if d_enc was, and then
    // the producer of d_enc has raised found_inf before this kernel started (what lets phase 2 apply the optimizer itself).
    // (the largest magnitude as a bit pattern: non-negative floats order like their patterns, and a NaN -- which fmaxf would
    //  silently drop -- sorts above infinity)
    uint32_t umax = 0u;
#ifdef STUB_UMAX   // dev stub: no maximum over the products, no wave fold
    umax = __float_as_uint(v[0]) & 0x7fffffffu;
#else
#pragma unroll
    for (int k = 0; k < 16; ++k) umax = max(umax, __float_as_uint(v[k]) & 0x7fffffffu);
#endif
    // A non-finite run raises the flag HERE, in phase 1: every block of phase 2 then reads a final found_inf before it applies the
    // optimizer to its slice (no partly stepped table), whoever produced d_enc.
    if (head && umax >= 0x7f800000u && b.found_inf) *b.found_inf = 1;
    const bool emit = head && umax != 0u && umax <= 0x7f800000u;     // (a NaN run is dropped; the step is skipped anyway)
    // (Round 6, measured and rejected: for the levels without run sums the largest magnitude is fl(max w * max |g|) -- rounding is
    //  monotone -- so the sixteen products can wait until the tile's exponent is known, w[c] * (g[k] * 2^e), and stay out of the
    //  registers that cross the barrier: 39 instructions fewer per level, phase 1 293 -> 309 us.  The stretch between the two barriers
    //  is the critical path of a level; what is added there costs more than what is saved in front of the first.)
    // Ranks within the slices.  The two x-neighbour corners of a cell (c, c + 1) almost always fall into the same slice (their
    // indices differ in the lowest bits: +1 in a dense level, ^1 for even x in a hashed one), so a PAIR takes one returning
    // atomic (+2) and, below, one 16-byte store.  (Measured: 325 -> 321 us only.  Stubbing out the atomics, the sorted stores or
    // the copy-out alone saves 114 / 125 / 96 us of 325, all three together 131: the three LDS phases are not additive costs but
    // alternatives on one critical path -- barrier to barrier -- and halving the operations of two of them moves little.)
    BT(2)
    uint32_t rk[8];   // (ranks of the four PAIRS: even entries)
    {
      // the tile's largest magnitude: folded over the wave with shuffles, ONE LDS atomic per wave.  (atomicMax from every lane is
      // rewritten by hipcc's atomic optimizer into a scalar loop over the active lanes -- ~6 scalar instructions per lane, 370 per
      // wave and level: half of this kernel's instruction stream, profiles/r03_pmc_sq_summary.json)
      uint32_t wm = emit ? umax : 0u;
#ifndef STUB_UMAX
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, o));
#endif
      if (lane == 0 && wm) atomicMax(&vmax_s[par], wm);
    }
#ifdef STUB_ATOM   // dev stub: no ranking atomics
#pragma unroll
    for (int c = 0; c < 8; ++c) rk[c] = (uint32_t)(tid * 8 + c) & 63u;
    if (false) {
#else
    if (emit) {
#endif
      // the four pair atomics go out back to back (no branch between them: one LDS round trip for all four instead of one each)
#pragma unroll
      for (int c = 0; c < 8; c += 2) rk[c] = atomicAdd(&cnt[par][idx[c] >> sl], 1u);
      // (a pair that straddles two slices -- BIN_CHUNK_WORDS -- takes a second record, in the slice of its second corner)
#pragma unroll
      for (int c = 0; c < 8; c += 2) if ((idx[c] >> sl) != (idx[c + 1] >> sl)) rk[c + 1] = atomicAdd(&cnt[par][idx[c + 1] >> sl], 1u);
    }
    BT(3)
    __syncthreads();
    BT(4)
    // Exclusive prefix over the slice counters, by EVERY wave for itself: one LDS read, six DPP adds (row shifts + the two row
    // broadcasts: vector-ALU speed), the bases parked in a wave-private LDS row for the per-corner lookups below.  Round 5 had wave 0
    // scan through six dependent ds_bpermute round trips (~1 000 ticks) with the other seven waves waiting at a second barrier.
    const uint32_t n_sl = cnt[par][lane];
    const uint32_t inc = wave_incl_scan(n_sl);
    base_w[tid >> 6][lane] = inc - n_sl;
    const uint32_t total_w = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const int ex_t = (int)((vmax_s[par] >> 23) & 0xFFu) - 127;          // floor(log2(max)); inf / huge -> large -> shift 0
    const int sh = min(max(14 - ex_t, BIN_MIN_SHIFT), BIN_MAX_SHIFT);
#ifndef STUB_DESC
    if (tid < 64) b.desc[((size_t)l * BIN_MAX_SLICES + lane) * b.ntiles + tile] = (inc - n_sl) | (n_sl << 13) | ((uint32_t)(sh + BIN_SHIFT_BIAS) << 27);
#endif
    BT(5)
    BT(6)
#ifdef STUB_STORE
    if (emit && v[0] == 12345.f) {
#else
    if (emit) {
#endif
      const float sc = __uint_as_float((uint32_t)(127 + sh) << 23);   // 2^shift
      const uint32_t* const bw = base_w[tid >> 6];
      uint32_t b0[4];
#pragma unroll
      for (int c = 0; c < 8; c += 2) b0[c >> 1] = bw[idx[c] >> sl];     // (all four slice bases requested before the first store)
      // record p of the chunk sits at word 3 p: the value pair at the 8-byte aligned one of words (3 p, 3 p + 1), the slot word beside it;
      // positions beyond the LDS tile (only with straddling pairs: BIN_CHUNK_WORDS) are written to the pool directly
      uint32_t* const chunk = b.pool + ((size_t)l * b.ntiles + tile) * BIN_CHUNK_WORDS;
      auto put = [&](uint32_t pos, uint32_t slots, uint32_t h0, uint32_t h1) __attribute__((always_inline)) {
        const uint32_t odd = pos & 1u;
        if (pos < (uint32_t)BIN_CHUNK) {   // (two branches, not a pointer select: that would be a flat store)
          uint32_t* const w = sorted + 3u * pos;
          *(uint2*)(w + odd) = make_uint2(h0, h1);
          w[odd ? 0 : 2] = slots;
        } else {
          uint32_t* const w = chunk + 3u * pos;
          *(uint2*)(w + odd) = make_uint2(h0, h1);
          w[odd ? 0 : 2] = slots;
        }
      };
#pragma unroll
      for (int c = 0; c < 8; c += 2) {
        const uint32_t hw0 = bin_pack_h2(v[2 * c] * sc, v[2 * c + 1] * sc), hw1 = bin_pack_h2(v[2 * c + 2] * sc, v[2 * c + 3] * sc);
        const uint32_t s0 = idx[c] >> sl, s1 = idx[c + 1] >> sl;
        const uint32_t m0 = idx[c] & slot_mask, m1 = idx[c + 1] & slot_mask;
        if (s0 == s1) put(b0[c >> 1] + rk[c], m0 | (m1 << 13), hw0, hw1);
        else {
          put(b0[c >> 1] + rk[c], m0 | (m0 << 13), hw0, 0u);
          put(bw[s1] + rk[c + 1], m1 | (m1 << 13), hw1, 0u);
        }
      }
    }
    if (k8 == 7 && l + 1 < p.level_hi) fetch_gw(l + 1);   // (behind the stores: the sixteen products are dead, the eight words fit the 80-register budget of three blocks per CU)
    BT(7)
    __syncthreads();
    BT(8)
    // (counters and maximum of this parity: next touched by the atomics of level l + 2, two barriers from here)
    if (tid < BIN_MAX_SLICES) cnt[par][tid] = 0;
    if (tid == 0) vmax_s[par] = 0u;
    cp_n = (min(total_w, (uint32_t)BIN_CHUNK) * (BIN_REC_WORDS * 4u) + 15u) / 16u;      // 16-byte pieces of the level's pair records
    cp_dst = (uint4*)(b.pool + ((size_t)l * b.ntiles + tile) * BIN_CHUNK_WORDS);
    BT(9)
  }
#pragma unroll
  for (int i = 0; i < (BIN_CHUNK * BIN_REC_WORDS * 4 / 16 + BIN_TILE - 1) / BIN_TILE; ++i) copy_piece(i);   // the last level's tile
  BT_FLUSH
}

// Optimizer fused into phase 2 (single-GPU training): the block that owns a slice holds its exact gradient sums in LDS, so it
// applies Adam to those entries itself -- the gradient never goes to HBM and comes back (16 of the 34 bytes per parameter the
// separate kernel moves).  Same arithmetic, in the same order, as k_adam (adam.hip); the step constants are derived from the
// optimizer state words as they stand before the step (k_adam, launched afterwards for the MLP blocks, advances them).
struct AccAdam {
  float* p; float* m; float* v; h16* t16;      // master parameters, moments, fp16 shadow of the table (NULL p = off)
  const int* si; const float* sf;              // optimizer state (adam.hip): si[2] found_inf, si[4] steps of block 0, sf[0] scale, sf[1] lr
  float lr, beta1, beta2, eps; double log_beta1, log_beta2;
};
struct AccParams {
  const uint32_t* pool; const uint32_t* desc; float* grad; int32_t* found_inf; AccAdam ad;
  h16* wire; float wire_mul;   // data parallelism with fp16 on the wire: the slice leaves as fp16(sum * wire_mul) -- no fp32 gradient
  int ntiles, level_lo, n_levels_here;
  uint32_t blk_start[ALN_MAX_LEVELS + 1];   // first block of each launched level (levels enumerated from level_lo)
  uint32_t size[ALN_MAX_LEVELS], offset[ALN_MAX_LEVELS];
  uint8_t slice_log2[ALN_MAX_LEVELS];
};

// fp16 record value -> 64-bit fixed point, exact and order-independent under integer addition.  A record value h carries the
// true value t = h * 2^-shift (shift = the tile's up-scaling, phase 1); the accumulator holds round(t * 2^U).  With the default
// U = 35 (2^-35 = the smallest fp16 denormal, 2^-24, scaled up by 2^11) the conversion of every record is exact.  Through double
// precision: h * 2^(U - shift) has at most 11 significant bits and lies below 2^51, so adding 1.5 * 2^52 parks it (rounded to an
// integer, ties to even, when U < 35 leaves fraction bits) in the low mantissa bits, two's complement
// (v_cvt_f64_f32, v_ldexp_f64, v_add_f64 and one 64-bit subtract; the add side of phase 2 is VALU-bound).
// (ds_add_f32 runs at 0.33 lanes/clk/CU on gfx950, ds_add_u64 at 4.6: profiles/r02_probe_lds_atomics.txt.)
__device__ inline long long fx_from_half_d(h16 h, int e) {
  const double z = ldexp((double)(float)h, e) + 6755399441055744.0;
  return __double_as_longlong(z) - 0x4338000000000000LL;
}
// the same number with the scaling done in single precision: h * 2^e (sc = 2^e, -126 <= e <= 127) is exact -- 11 significant bits, far
// inside the exponent range -- and a full-rate multiply, where v_ldexp_f64 is not
__device__ inline long long fx_from_half_s(h16 h, float sc) {
  const double z = (double)((float)h * sc) + 6755399441055744.0;
  return __double_as_longlong(z) - 0x4338000000000000LL;
}
#define FX_UNIT_LOG2 (24 + BIN_MAX_SHIFT)   // 35

#ifdef ACC_TIMING   // dev builds only: shader-clock ticks per section of phase 2, block ACC_TIMING_BLOCK, waves 0 and 15
__device__ long long g_acc_t[2][12];
extern "C" int aln_debug_read_acc_timing(long long* host_out, int reset) {
  if (reset) { long long z[24] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_acc_t), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_acc_t), sizeof(long long) * 24);
}
#ifndef ACC_TIMING_BLOCK
#define ACC_TIMING_BLOCK 700
#endif
#define AT_DECL long long at_acc[12] = {0}; long long at_last = clock64();
#define AT(i) { long long at_now = clock64(); at_acc[i] += at_now - at_last; at_last = at_now; }
#define AT_FLUSH if (blockIdx.x == ACC_TIMING_BLOCK && (tid == 0 || tid == 960)) { for (int i = 0; i < 12; ++i) g_acc_t[tid != 0][i] += at_acc[i]; }
#else
#define AT_DECL
#define AT(i)
#define AT_FLUSH
#endif
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3v __attribute__((ext_vector_type(3), aligned(4)));   // a 12-byte pair record
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_max_u16(uint32_t a, uint32_t b) {   // v_pk_max_u16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2v, a), __builtin_bit_cast(u16x2v, b)));
}
#define ACC_THREADS 1024
#ifndef ACC_RB
#define ACC_RB 8    // tiles (= runs of the slice) per batch of the accumulate loop
#endif
typedef uint32_t acc_sv __attribute__((ext_vector_type(ACC_RB)));
#ifndef ACC_TR
#define ACC_TR 4    // trips of 64 pair records per batch whose loads are requested a batch ahead (two batches of registers)
#endif
__global__ __launch_bounds__(ACC_THREADS) void k_encode_bwd_accum(AccParams a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char acc_smem[];
  long long* acc = (long long*)acc_smem;   // [2 features][slice entries] <= 128 KB: one plane per feature -- with the two features of
                                           // an entry side by side, a wave's 64 slots fell onto 16 bank quads (PMC: 65 % of the
                                           // LDS cycles of this kernel were bank conflicts); a plane spreads them over 32 bank pairs
  __shared__ unsigned long long bound_s;
  __shared__ int smin_s;
  __shared__ uint32_t next_batch_s;   // next batch of ACC_RB tiles nobody has taken yet
  __shared__ float adam_c[4];   // skip, 1 / loss scale, lr / bc1, 1 / sqrt(bc2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = ACC_THREADS / 64;
  AT_DECL
  int li = 0;
  while (li + 1 < a.n_levels_here && blockIdx.x >= a.blk_start[li + 1]) ++li;
  const int l = a.level_lo + li;
  const uint32_t sl = a.slice_log2[l];
  const int s = blockIdx.x - a.blk_start[li];
  const uint32_t e0 = (uint32_t)s << sl;
  const uint32_t ne = min(1u << sl, a.size[l] - e0);   // entries of this slice (the LDS accumulators beyond them are never touched)
  const uint32_t* d = a.desc + ((size_t)l * BIN_MAX_SLICES + s) * a.ntiles;
  // Range guard.  A record is below 2^(15 - shift) in magnitude (phase 1 scales every tile's largest value into [2^14, 2^15), up or
  // down), so the sum over the runs of count * that bounds any accumulator of the slice; if that bound times 2^35 could pass 2^62, the block accumulates in a coarser unit 2^-U
  // (records are then ROUNDED to it -- still one fixed integer per record, so the sums stay order-independent).  In training
  // this never triggers for the hashed levels; it is what lets the two coarsest levels (thousands of records per entry at the
  // largest loss scales) share the exact path.
  const uint32_t plane = 1u << sl;   // entries per plane
  if (tid == 0) { bound_s = 0ull; smin_s = 0; next_batch_s = 0u; }
  if (a.ad.p && tid == 64) {   // (a lane of wave 1: off the critical path, needed at the flush only)
    const int found = a.ad.si[2], t = a.ad.si[4] + 1;
    const float lr = a.ad.sf[1] > 0.f ? a.ad.sf[1] : a.ad.lr;
    const double bc1 = 1.0 - exp((double)t * a.ad.log_beta1), bc2 = 1.0 - exp((double)t * a.ad.log_beta2);
    adam_c[0] = found ? 1.f : 0.f; adam_c[1] = 1.0f / a.ad.sf[0];
    adam_c[2] = (float)((double)lr / bc1); adam_c[3] = (float)(1.0 / sqrt(bc2));
  }
  const uint32_t* pool = a.pool + (size_t)l * a.ntiles * BIN_CHUNK_WORDS;
  // (the first descriptors of the bound pass are requested before the accumulators are cleared: one HBM round trip under the LDS stores)
  const uint32_t qd0 = tid < a.ntiles ? d[tid] : 0u, qd1 = tid + ACC_THREADS < a.ntiles ? d[tid + ACC_THREADS] : 0u;
  for (uint32_t i = tid; i < ne; i += ACC_THREADS) { acc[i] = 0ll; acc[plane + i] = 0ll; }
  AT(0)
  __syncthreads();
  AT(1)
  {
    unsigned long long bsum = 0ull; int smin = 0;
    auto fold = [&](uint32_t q) __attribute__((always_inline)) {
      const int sh = DESC_SHIFT(q);
      if (DESC_COUNT(q)) { bsum += (unsigned long long)DESC_COUNT(q) << (15 - sh); smin = min(smin, sh); }   // |record| < 2^(15 - shift)
    };
    fold(qd0); fold(qd1);
    for (int t = tid + 2 * ACC_THREADS; t < a.ntiles; t += ACC_THREADS) fold(d[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bsum += __shfl_xor(bsum, o); smin = min(smin, __shfl_xor(smin, o)); }
    if (lane == 0 && bsum) { atomicAdd(&bound_s, bsum); atomicMin(&smin_s, smin); }
  }
  AT(2)
  __syncthreads();
  AT(3)
  int U = FX_UNIT_LOG2;
  // (a tile that was scaled DOWN carries records up to 2^(16 - shift): the double-precision conversion needs them below 2^51 units)
  { const unsigned long long bd = bound_s; const int lg = bd ? 64 - __clzll(bd) : 0; U = min(FX_UNIT_LOG2 + smin_s, 62 - lg); }   // bd < 2^lg
#ifdef STUB_ACC_NOATOM
  long long stub_x = 0;
#endif
  uint32_t badbits = 0u;   // largest |half| seen, as bit patterns (two packed 15-bit maxima): >= 0x7C00 = a record value was inf or nan
  auto add = [&](u32x3v w, uint32_t odd_pos, int ex) {   // pair record at an odd / even position of its chunk; ex = U - shift of the run's tile
    // (an odd record keeps its slot word first, an even one last: BIN_REC_WORDS)
    const bool odd = odd_pos != 0u;
    const uint32_t slots = odd ? w.x : w.z, v0 = odd ? w.y : w.x, v1 = odd ? w.z : w.y;
    const h16x2 h0 = __builtin_bit_cast(h16x2, v0), h1 = __builtin_bit_cast(h16x2, v1);
    badbits = pk_max_u16(badbits, pk_max_u16(v0 & 0x7FFF7FFFu, v1 & 0x7FFF7FFFu));
    const uint32_t s0 = slots & 0x1FFFu, s1 = (slots >> 13) & 0x1FFFu;
#if defined(STUB_ACC_NOATOM)   // dev stubs: conversions without the LDS atomics / atomics without the conversions
    const float sc = __uint_as_float((uint32_t)(127 + ex) << 23);
    long long q = fx_from_half_s(h0[0], sc) ^ fx_from_half_s(h0[1], sc);
    if (v1 & 0x7FFF7FFFu) q ^= fx_from_half_s(h1[0], sc) ^ fx_from_half_s(h1[1], sc);
    stub_x ^= q + s0 + s1;
#elif defined(STUB_ACC_NOCONV)
    atomicAdd((unsigned long long*)&acc[s0], (unsigned long long)v0);
    atomicAdd((unsigned long long*)&acc[plane + s0], (unsigned long long)(v0 >> 16));
    if (v1 & 0x7FFF7FFFu) {
      atomicAdd((unsigned long long*)&acc[s1], (unsigned long long)v1);
      atomicAdd((unsigned long long*)&acc[plane + s1], (unsigned long long)(v1 >> 16));
    }
#elif defined(ACC_LDEXP64)
    atomicAdd((unsigned long long*)&acc[s0], (unsigned long long)fx_from_half_d(h0[0], ex));
    atomicAdd((unsigned long long*)&acc[plane + s0], (unsigned long long)fx_from_half_d(h0[1], ex));
    if (v1 & 0x7FFF7FFFu) {   // (the empty second half of a pair that straddled two slices adds nothing)
      atomicAdd((unsigned long long*)&acc[s1], (unsigned long long)fx_from_half_d(h1[0], ex));
      atomicAdd((unsigned long long*)&acc[plane + s1], (unsigned long long)fx_from_half_d(h1[1], ex));
    }
#else
    const float sc = __uint_as_float((uint32_t)(127 + ex) << 23);   // 2^ex
    atomicAdd((unsigned long long*)&acc[s0], (unsigned long long)fx_from_half_s(h0[0], sc));
    atomicAdd((unsigned long long*)&acc[plane + s0], (unsigned long long)fx_from_half_s(h0[1], sc));
    if (v1 & 0x7FFF7FFFu) {   // (the empty second half of a pair that straddled two slices adds nothing)
      atomicAdd((unsigned long long*)&acc[s1], (unsigned long long)fx_from_half_s(h1[0], sc));
      atomicAdd((unsigned long long*)&acc[plane + s1], (unsigned long long)fx_from_half_s(h1[1], sc));
    }
#endif
  };
  {
    // Batches of ACC_RB consecutive tiles are handed out DYNAMICALLY (one LDS counter per block).  With equal static shares the 16
    // waves of a block finished up to 65 % apart (section clocks of a -DACC_TIMING build: loop times 99 K .. 164 K ticks per wave for
    // the same number of records -- whoever loses the issue arbitration of its SIMD falls behind for good), and the early ones sat at
    // the block's barrier for a quarter of the kernel.  Integer accumulation is order-independent, so who adds which run changes no bit.
    // Pipeline per wave, no drain anywhere: grab + descriptor load two batches ahead (ACC_RB lanes, one 4-byte word each), record loads
    // one batch ahead (12 bytes per lane and trip), consume.  The batch index comes back through readfirstlane.
    const int nbatch = (a.ntiles + ACC_RB - 1) / ACC_RB;
    // (the counter's returning atomic is ISSUED one stage before its result is read: an LDS round trip behind the other waves' accumulate
    //  traffic is ~1000 ticks, as long as the requests of a whole batch take to issue)
    auto grab_issue = [&]() __attribute__((always_inline)) {
      uint32_t g = 0u;
      if (lane == 0) g = atomicAdd(&next_batch_s, 1u);
      return g;
    };
    auto grabbed = [&](uint32_t g) __attribute__((always_inline)) { return (int)__builtin_amdgcn_readfirstlane(g); };
    // (every load of the pipeline is UNCONDITIONAL -- out-of-range lanes read a valid address and drop the value: behind a load in a
    //  branch hipcc's wait insertion no longer knows how many younger loads are in flight and waits for all of them, vmcnt(0), which
    //  put the batch requested a moment ago in front of every accumulate)
    auto load_desc = [&](int bt) __attribute__((always_inline)) {
      const int t = min(bt, nbatch - 1) * ACC_RB + (lane % ACC_RB);
      const uint32_t q = d[min(t, a.ntiles - 1)];
      return (bt < nbatch && lane < ACC_RB && t < a.ntiles) ? q : 0u;   // (0: start 0, count 0 -- nothing is requested for it)
    };
    // The runs of a batch -- ACC_RB tiles, ~32 pair records each at a hashed level -- are walked as ONE stream, 64 pair records per trip:
    // lane i of trip t takes record g = 64 t + i of the concatenation.  (One run per trip left half the lanes idle: the add side is
    // bound by instruction issue, and the conversions of a half-empty wave cost what those of a full one do.)  The descriptor lanes
    // (0 .. ACC_RB - 1) scan their counts (DPP) and pack, per run, the word offset of the stream's position 0 and U - shift; both go to
    // scalar registers, and a lane picks its run's fields with a compare + select chain over them (a bpermute would queue behind the
    // accumulate atomics of the batch before: LDS instructions complete in order).
    // The first ACC_TR trips of a batch are requested a batch ahead (registers); a longer batch loops.
    u32x3v r[2][ACC_TR]; uint32_t pe[2][ACC_TR], inc_[2], fld_[2], tot[2];
    // pe: word offset of the lane's record from the batch's first chunk (20 bits; its parity = the record's) | (U - shift + 128) << 20
    // (vector VALUES, not arrays: through an array hipcc turned the select chain into a select of addresses and one scratch load)
    auto locate = [&](uint32_t g, acc_sv sI, acc_sv sF) __attribute__((always_inline)) {
      uint32_t f = sF[0];
#pragma unroll
      for (int u = 1; u < ACC_RB; ++u) f = g >= sI[u - 1] ? sF[u] : f;
      return ((3u * g + f) & 0xFFFFFu) | (f & 0xFFF00000u);
    };
    auto spread = [&](uint32_t x) __attribute__((always_inline)) {   // descriptor lanes -> scalar registers
      acc_sv s;
#pragma unroll
      for (int u = 0; u < ACC_RB; ++u) s[u] = (uint32_t)__builtin_amdgcn_readlane((int)x, u);
      return s;
    };
    auto request = [&](int buf, int bt, uint32_t dq) __attribute__((always_inline)) {
      const uint32_t nq = DESC_COUNT(dq);   // (0 beyond the ACC_RB descriptor lanes)
      uint32_t inc = nq;
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true);    // row_shr:1
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true);    // row_shr:2
      inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true);    // row_shr:4
      if (ACC_RB > 8) inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true);    // row_shr:8
      const uint32_t fld = (((uint32_t)lane * BIN_CHUNK_WORDS + 3u * (DESC_START(dq) - (inc - nq))) & 0xFFFFFu) | ((uint32_t)(U - DESC_SHIFT(dq) + 128) << 20);
      inc_[buf] = inc; fld_[buf] = fld;
      const acc_sv sI = spread(inc), sF = spread(fld);
      tot[buf] = sI[ACC_RB - 1];
      const uint32_t* const src = pool + (size_t)min(bt, nbatch - 1) * (ACC_RB * BIN_CHUNK_WORDS);
#pragma unroll
      for (int t = 0; t < ACC_TR; ++t) {
        pe[buf][t] = locate(64u * t + (uint32_t)lane, sI, sF);
        r[buf][t] = *(const u32x3v*)(src + (64u * t + (uint32_t)lane < tot[buf] ? pe[buf][t] & 0xFFFFFu : 0u));   // (lanes past the stream: the batch's first words, unused)
      }
    };
    auto consume = [&](int buf, int bt) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < ACC_TR; ++t)
        if (64u * t + (uint32_t)lane < tot[buf]) add(r[buf][t], pe[buf][t] & 1u, (int)(pe[buf][t] >> 20) - 128);
      if (64u * ACC_TR < tot[buf]) {   // a long batch
        const uint32_t* const src = pool + (size_t)bt * (ACC_RB * BIN_CHUNK_WORDS);
        const acc_sv sI = spread(inc_[buf]), sF = spread(fld_[buf]);
        for (uint32_t g = 64u * ACC_TR + (uint32_t)lane; g < tot[buf]; g += 64u) {
          const uint32_t q = locate(g, sI, sF);
          add(*(const u32x3v*)(src + (q & 0xFFFFFu)), q & 1u, (int)(q >> 20) - 128);
        }
      }
    };
    AT(4)
    const uint32_t g0 = grab_issue(), g1 = grab_issue();
    uint32_t gC = grab_issue();
    int bA = grabbed(g0), bB = grabbed(g1);
    uint32_t dA = load_desc(bA), dB = load_desc(bB);
    request(0, bA, dA);
    while (true) {   // here: batch A's records are in flight (buffer 0), batch B has its descriptors, batch C's index is on its way
      const int bC = grabbed(gC);
      const uint32_t gD = grab_issue();
      const uint32_t dC = load_desc(bC);
      request(1, bB, dB);
      AT(5)
      consume(0, bA);
      AT(7)
      if (bB >= nbatch) break;
      const int bD = grabbed(gD);
      gC = grab_issue();
      const uint32_t dD = load_desc(bD);
      request(0, bC, dC);
      AT(5)
      consume(1, bB);
      AT(7)
      if (bC >= nbatch) break;
      bA = bC; bB = bD; dB = dD;
    }
  }
#ifdef STUB_ACC_NOATOM
  if (stub_x == 0x123456789LL) acc[tid] = stub_x;
#endif
  const bool bad = (badbits & 0xFFFFu) >= 0x7C00u || (badbits >> 16) >= 0x7C00u;
  AT(8)
  __syncthreads();
  AT(9)
  const size_t g0 = 2 * ((size_t)a.offset[l] + e0);
  const double unit = ldexp(1.0, -U);
  if (a.ad.p) {
    // optimizer step for the entries of this slice: both features of an entry per lane (8-byte accesses), every entry -- torch's Adam
    // also moves a parameter whose gradient is zero while its moments are not.  A non-finite record implies that d_enc was
    // non-finite, i.e. si[2] was raised before this launch and every block skips alike.
    const bool skip = adam_c[0] != 0.f;
    const float inv_scale = adam_c[1], step_size = adam_c[2], inv_sqrt_bc2 = adam_c[3];
    const float b1 = a.ad.beta1, b2 = a.ad.beta2, c1 = 1.f - a.ad.beta1, c2 = 1.f - a.ad.beta2, eps = a.ad.eps;
    if (!skip && !bad) {   // (bad without skip: a caller fed non-finite records without raising the flag -- never write them into the parameters)
      float2* const P2 = (float2*)(a.ad.p + g0); float2* const M2 = (float2*)(a.ad.m + g0); float2* const V2 = (float2*)(a.ad.v + g0);
      h16x2* const T2 = (h16x2*)(a.ad.t16 + g0);
      // (a full slice is ACC_EPT entries per thread: their operands are all requested before the first is used -- as a plain loop
      //  hipcc waits for every entry's three loads in turn, eight HBM round trips in a row at the end of every block)
      constexpr int ACC_EPT = BIN_SLICE / ACC_THREADS;
      for (uint32_t eb = 0; eb < ne; eb += ACC_EPT * ACC_THREADS) {
        float2 p2[ACC_EPT], m2[ACC_EPT], v2[ACC_EPT];
#pragma unroll
        for (int i = 0; i < ACC_EPT; ++i) {
          const uint32_t e = eb + tid + i * ACC_THREADS;
          if (e < ne) { p2[i] = P2[e]; m2[i] = M2[e]; v2[i] = V2[e]; }
        }
#pragma unroll
        for (int i = 0; i < ACC_EPT; ++i) {
          const uint32_t e = eb + tid + i * ACC_THREADS;
          if (e >= ne) continue;
          float gg[2] = {(float)((double)acc[e] * unit), (float)((double)acc[plane + e] * unit)};
          float pp[2] = {p2[i].x, p2[i].y}, mm[2] = {m2[i].x, m2[i].y}, vv[2] = {v2[i].x, v2[i].y};
          h16x2 t2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float gi = gg[k] * inv_scale;
            const float mi = b1 * mm[k] + c1 * gi;
            const float vi = b2 * vv[k] + c2 * gi * gi;
            mm[k] = mi; vv[k] = vi;
            const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
            pp[k] -= step_size * (mi / denom);
            t2[k] = (h16)pp[k];
          }
          M2[e] = make_float2(mm[0], mm[1]); V2[e] = make_float2(vv[0], vv[1]); P2[e] = make_float2(pp[0], pp[1]); T2[e] = t2;
        }
      }
    }
    if (bad && a.found_inf) *a.found_inf = 1;   // (already set by the producer of d_enc; kept for a caller that feeds records of its own)
    AT(10)
    AT_FLUSH
    return;
  }
  if (a.wire) {
    // The payload of the gradient exchange written here: exactly what aln_grad_pack_f16 makes of the fp32 gradient this block would
    // have added to a zeroed table -- fp16(float(sum) * 1 / world), zeros included (the wire buffer is not cleared between steps) --
    // without the fp32 table's read-modify-write and without the packing pass (57 + 57 + 57 + 28 MB of HBM traffic per step).
    h16x2* const w2 = (h16x2*)(a.wire + g0);
    for (uint32_t e = tid; e < ne; e += ACC_THREADS) {
      const long long q0 = acc[e], q1 = acc[plane + e];
      const float f0 = q0 != 0ll ? (float)((double)q0 * unit) : 0.f, f1 = q1 != 0ll ? (float)((double)q1 * unit) : 0.f;
      h16x2 o; o[0] = (h16)(f0 * a.wire_mul); o[1] = (h16)(f1 * a.wire_mul);
      w2[e] = o;
    }
    // a non-finite record anywhere in the block poisons the slice's first element (thread 0 wrote it above: program order), so the
    // post-reduction watch (aln_grad_unpack_f16) sees it on EVERY rank whether or not the caller passed a flag
    if (__syncthreads_or(bad) && tid == 0) {
      a.wire[g0] = (h16)__builtin_nanf("");
      if (a.found_inf) *a.found_inf = 1;   // (the engine reduces this flag over the ranks: every rank skips the step)
    }
    return;
  }
  float* g = a.grad + g0;
  for (uint32_t i = tid; i < 2 * ne; i += ACC_THREADS) {   // g is [entry][feature]
    const long long q = acc[(i & 1u) * plane + (i >> 1)];
    if (q != 0ll) g[i] += (float)((double)q * unit);    // one rounding of the exact sum; no other block owns this entry
  }
  if (bad) {   // poison the slice (torch's GradScaler looks at the gradient tensor itself) and raise the engine's flag
    g[0] = __builtin_nanf("");
    if (a.found_inf) *a.found_inf = 1;
  }
}

extern "C" int32_t aln_encode_bwd_binned_tile_rows(void) { return BIN_TILE; }
extern "C" int64_t aln_encode_bwd_binned_ws_bytes(const AlnEncDesc* e, int32_t rows) {
  if (!e || !e->use_grid || rows <= 0) return 0;
  const int64_t ntiles = (rows + BIN_TILE - 1) / BIN_TILE, nl = e->grid.n_levels;
  // [pool: chunks of pair records][descriptors]
  return nl * ntiles * (int64_t)(BIN_CHUNK_WORDS * sizeof(uint32_t)) + nl * BIN_MAX_SLICES * ntiles * (int64_t)sizeof(uint32_t);
}

static int binned_launch(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                         const float* xyz, int32_t rows, int32_t rows_pass1, int32_t stride1, int32_t stride2,
                         const uint16_t* perm, const void* d_enc, float* grad_table, void* ws, int32_t level_lo,
                         int32_t level_hi, int32_t* found_inf, const AlnAdamFuse* adam, void* wire_f16, float wire_mul, int phases, void* stream) {
  BinParams b;
  ALN_REQUIRE(phases >= 1 && phases <= 3 && (phases == 3 || !adam), "encode_bwd_binned: phases must be 1 (records), 2 (accumulate) or 3 (both)");
  if (int rc = fill_params(b.p, e, nullptr, rays_o, rays_d, z, xyz, rows, stride1)) return rc;
  // (with `adam` the caller's aln_adam_step(skip_grid = 1) advances the table's step counter: an empty launch must not pass for a step)
  ALN_REQUIRE(!adam || (rows > 0 && e->use_grid && level_lo < level_hi), "encode_bwd_binned: the fused optimizer step needs rows and levels; "
              "run aln_adam_step without skip_grid for an empty batch");
  if (!e->use_grid) return 0;
  ALN_REQUIRE(0 <= level_lo && level_lo <= level_hi && level_hi <= (int)e->grid.n_levels, "encode_bwd_binned: level range [%d, %d)",
              level_lo, level_hi);
  if (rows == 0) {   // nothing to scatter: an fp32 table keeps its zeros, the wire payload of these levels has to be written as zeros
    if (wire_f16 && level_lo < level_hi && (phases & 2)) {
      const size_t lo = 2 * (size_t)e->grid.offset[level_lo];
      const size_t hi = level_hi < (int)e->grid.n_levels ? 2 * (size_t)e->grid.offset[level_hi] : 2 * ((size_t)e->grid.offset[level_hi - 1] + e->grid.size[level_hi - 1]);
      ALN_REQUIRE(hipMemsetAsync((h16*)wire_f16 + lo, 0, (hi - lo) * sizeof(h16), (hipStream_t)stream) == hipSuccess, "encode_bwd_binned: memset of the wire buffer failed");
    }
    return 0;
  }
  ALN_REQUIRE(d_enc && (phases == 1 || grad_table || adam || wire_f16) && ws, "encode_bwd_binned: NULL pointer");
  ALN_REQUIRE(!wire_f16 || (!adam && ((uintptr_t)wire_f16 & 3) == 0 && e->grid.n_features == 2), "encode_bwd_binned: the fp16 wire output excludes the fused "
              "optimizer and needs a 4-byte aligned buffer");
  ALN_REQUIRE(!adam || (adam->params && adam->m && adam->v && adam->table_f16 && adam->state_i && adam->state_f && e->grid.n_features == 2),
              "encode_bwd_binned: incomplete optimizer descriptor");
  ALN_REQUIRE(!adam || ((((uintptr_t)adam->params | (uintptr_t)adam->m | (uintptr_t)adam->v) & 7) == 0 && ((uintptr_t)adam->table_f16 & 3) == 0),
              "encode_bwd_binned: optimizer buffers must be 8-byte aligned (fp16 shadow: 4)");
  ALN_REQUIRE(0 <= rows_pass1 && rows_pass1 <= rows && (rows_pass1 == rows || stride2 > 0), "encode_bwd_binned: bad pass split");
  ALN_REQUIRE(!perm || (!xyz && stride1 > 0 && rows_pass1 % stride1 == 0 && rows_pass1 < rows &&
                        (int64_t)(rows_pass1 / stride1) * (stride1 + stride2) == rows),
              "encode_bwd_binned: a depth order needs the two-pass row layout (rays x stride1, then rays x stride2)");
  ALN_REQUIRE(e->grid.log2_hashmap_size <= BIN_SLICE_LOG2 + 6, "encode_bwd_binned: tables above 2^19 entries are not supported");
  if (level_lo == level_hi) return 0;
  const int ntiles = (rows + BIN_TILE - 1) / BIN_TILE, nl = e->grid.n_levels;
  b.p.level_lo = level_lo; b.p.level_hi = level_hi;
  b.d_enc = (const h16*)d_enc; b.found_inf = found_inf; b.ntiles = ntiles; b.rows1 = rows_pass1; b.stride2 = stride2 > 0 ? stride2 : 1;
  b.pool = (uint32_t*)ws; b.perm = perm;
  b.desc = (uint32_t*)((char*)ws + (size_t)nl * ntiles * (BIN_CHUNK_WORDS * sizeof(uint32_t)));
  AccParams a;
  for (int l = 0; l < ALN_MAX_LEVELS / 4; ++l) b.slice_log2_w[l] = 0u;
  for (int l = 0; l < ALN_MAX_LEVELS; ++l) {
    a.slice_log2[l] = (uint8_t)(l < nl ? bin_slice_log2(e->grid.size[l]) : 0);
    b.slice_log2_w[l >> 2] |= (uint32_t)a.slice_log2[l] << (8 * (l & 3));
  }
  if (phases & 1) {
    hipLaunchKernelGGL(k_encode_bwd_bin, dim3(ntiles), dim3(BIN_TILE), 0, (hipStream_t)stream, b);
    ALN_CHECK_LAUNCH("encode_bwd_bin");
  }
  if (!(phases & 2)) return 0;
  a.pool = b.pool; a.desc = b.desc; a.grad = grad_table; a.found_inf = found_inf; a.ntiles = ntiles;
  a.wire = (h16*)wire_f16; a.wire_mul = wire_mul;
  a.ad = AccAdam{};
  if (adam) a.ad = AccAdam{adam->params, adam->m, adam->v, (h16*)adam->table_f16, adam->state_i, adam->state_f, adam->lr, adam->beta1,
                           adam->beta2, adam->eps, log((double)adam->beta1), log((double)adam->beta2)};
  a.level_lo = level_lo; a.n_levels_here = level_hi - level_lo;
  uint32_t nblk = 0, max_entries = 1;
  for (int l = level_lo; l < level_hi; ++l) {
    const uint32_t per = 1u << a.slice_log2[l], nsl = (e->grid.size[l] + per - 1) / per;
    ALN_REQUIRE(nsl <= BIN_MAX_SLICES, "encode_bwd_binned: level %d has %u slices", l, nsl);
    a.blk_start[l - level_lo] = nblk;
    nblk += nsl;
    if (per > max_entries) max_entries = per;
  }
  a.blk_start[level_hi - level_lo] = nblk;
  for (int l = 0; l < nl; ++l) { a.size[l] = e->grid.size[l]; a.offset[l] = e->grid.offset[l]; }
  static const bool lds_ok = hipFuncSetAttribute((const void*)k_encode_bwd_accum, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 2 * BIN_SLICE * (int)sizeof(long long)) == hipSuccess;
  ALN_REQUIRE(lds_ok, "encode_bwd_binned: cannot reserve 128 KB of LDS");
  // LDS for the largest slice of this launch (a launch of coarse levels only -- the last level group of the data-parallel
  // schedule -- then fits several blocks per CU)
  hipLaunchKernelGGL(k_encode_bwd_accum, dim3(nblk), dim3(ACC_THREADS), 2 * (size_t)max_entries * sizeof(long long), (hipStream_t)stream, a);
  ALN_CHECK_LAUNCH("encode_bwd_accum");
  return 0;
}
extern "C" int aln_encode_bwd_binned(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                                     const float* xyz, int32_t rows, int32_t rows_pass1, int32_t stride1, int32_t stride2,
                                     const uint16_t* perm, const void* d_enc, float* grad_table, void* ws, int32_t level_lo,
                                     int32_t level_hi, int32_t* found_inf, const AlnAdamFuse* adam, void* stream) {
  return binned_launch(e, rays_o, rays_d, z, xyz, rows, rows_pass1, stride1, stride2, perm, d_enc, grad_table, ws, level_lo, level_hi, found_inf,
                       adam, nullptr, 0.f, 3, stream);
}
// The same scatter with the gradient of levels [level_lo, level_hi) leaving as the fp16 PAYLOAD of the data-parallel exchange:
// wire_f16[2 * (offset[l] + entry) + feature] = fp16(float(sum) * wire_mul) for EVERY entry of those levels (zeros included) -- bit for bit
// what aln_grad_pack_f16(grad_table, mul = wire_mul) produces after aln_encode_bwd_binned into a zeroed grad_table; nothing is read
// from or written to an fp32 gradient table.  A non-finite record poisons the slice's first element and raises found_inf.
extern "C" int aln_encode_bwd_binned_wire(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                                          const float* xyz, int32_t rows, int32_t rows_pass1, int32_t stride1, int32_t stride2,
                                          const uint16_t* perm, const void* d_enc, void* ws, int32_t level_lo, int32_t level_hi,
                                          int32_t* found_inf, void* wire_f16, float wire_mul, void* stream) {
  ALN_REQUIRE(wire_f16, "encode_bwd_binned_wire: NULL wire buffer");
  return binned_launch(e, rays_o, rays_d, z, xyz, rows, rows_pass1, stride1, stride2, perm, d_enc, nullptr, ws, level_lo, level_hi, found_inf,
                       nullptr, wire_f16, wire_mul, 3, stream);
}
// The two phases on their own (the overlapped data-parallel exchange: phase 1 ONCE for all levels -- one launch instead of one per bucket,
// 270 instead of 348 us at the bench's batch -- then phase 2 bucket by bucket, each bucket's exchange starting behind its own launch).
// phases = 1: records of levels [level_lo, level_hi) into ws (grad_table / wire_f16 unused); phases = 2: accumulate those levels from
// ws into grad_table (added) or, with wire_f16, into the fp16 payload; phases = 3: both, as aln_encode_bwd_binned / _wire.  The same
// kernels on the same records: results are bit-identical however the levels are grouped.
extern "C" int aln_encode_bwd_binned_phase(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z,
                                           const float* xyz, int32_t rows, int32_t rows_pass1, int32_t stride1, int32_t stride2,
                                           const uint16_t* perm, const void* d_enc, float* grad_table, void* ws, int32_t level_lo,
                                           int32_t level_hi, int32_t* found_inf, void* wire_f16, float wire_mul, int32_t phases, void* stream) {
  return binned_launch(e, rays_o, rays_d, z, xyz, rows, rows_pass1, stride1, stride2, perm, d_enc, grad_table, ws, level_lo, level_hi, found_inf,
                       nullptr, wire_f16, wire_mul, phases, stream);
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
