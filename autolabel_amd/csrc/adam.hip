// Fused Adam + grad unscale + inf-skip + fp16 shadow refresh + grad zeroing, and the GradScaler state update.
// Replaces torch.optim.Adam (scripts/train.py:50-63: lr 5e-3, betas (0.9,0.99), eps 1e-15, weight_decay 1e-6
// on the MLP group only) and torch.cuda.amp.GradScaler step/update (autolabel/trainer.py:45-48).
// Spec: oracle/nerf_oracle.py:adam_update.  HBM-bound streaming kernel over the flat parameter buffer
// [grid table | MLP weights]: 16 B read + 18 B written per parameter.
#include "common.h"
#include <math.h>

// Parameter blocks follow torch's per-tensor semantics: a block whose gradient is None in the reference (the semantic heads
// when a batch has no labelled ray and no feature loss: autolabel/trainer.py:61-62,80-92) is skipped entirely -- no moment
// decay, no weight decay, and its private step counter (bias correction) does not advance.
//
// state (device):
//   si[0] = applied optimizer steps   si[1] = growth tracker   si[2] = found_inf flag   si[3] = scatter flag of the data-parallel
//   engine   si[4+b] = step count of block b   si[15] = arrival ticket of k_adam
//   sf[0] = loss scale   sf[1] = learning rate (> 0: overrides the host argument, so a captured hipGraph follows the scheduler)
// consts (optional, written by the last block for inspection): c[0] = skip all (0/1), c[3] = 1/scale_used, c[4+2b] = lr / bc1_b
//                                     (0 = block inactive), c[5+2b] = 1/sqrt(bc2_b)
#define ADAM_MAX_BLOCKS 8
struct AdamHyper { float lr, beta1, beta2, eps, wd_net, growth, backoff; int growth_interval; double log_beta1, log_beta2; };
struct AdamBlocks { int n; long long end[ADAM_MAX_BLOCKS]; int needs_sem[ADAM_MAX_BLOCKS]; int needs_sem_or_feat[ADAM_MAX_BLOCKS]; int feat_on;
                    int skip_grid; };   // skip_grid: block 0 was updated by phase 2 of the hash-grid scatter (encode.hip, AccAdam)
// The slices [lo, hi) of the hash table this launch owns (multiples of 4 parameters).  One slice [0, n_grid) on one GPU; under
// data parallelism with a sharded optimizer (aln_adam_step_ranges) every rank owns 1 / world of every gradient bucket, and its
// moment buffers m, v hold the owned slices back to back (slice k at mv[k]), then the MLP block at mlp_mv: the optimizer state
// of the table is 1 / world of the replicated one.
#define ADAM_MAX_RANGES 8
struct AdamRanges { int n; long long lo[ADAM_MAX_RANGES], hi[ADAM_MAX_RANGES], mv[ADAM_MAX_RANGES]; long long mlp_mv; };

// The step constants (skip flag, 1 / loss scale, per-block step size and bias correction) are a pure function of the state words
// as they stand BEFORE this step; every block derives them itself (one thread, a handful of pow() in double) instead of waiting
// for a one-thread prepare kernel.  The state itself (step counters, loss scale, growth tracker, found_inf reset) is advanced by
// the LAST block to finish (arrival ticket in si[15], self-resetting): by then every block has read the old state.
struct AdamConsts { float skip, inv_scale, step_size[ADAM_MAX_BLOCKS], inv_sqrt_bc2[ADAM_MAX_BLOCKS]; };
__device__ inline bool adam_block_active(const AdamBlocks& blk, int b, bool has_sem) {
  if (blk.needs_sem[b]) return has_sem;
  if (blk.needs_sem_or_feat[b]) return has_sem || blk.feat_on;
  return true;
}

__global__ void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       h16* __restrict__ table16, size_t n_grid, size_t n_total, int* si, float* sf, float* c_out, AdamHyper h,
                       AdamBlocks blk, AdamRanges rg, const int* counts, uint32_t* step_dev, const h16* __restrict__ gwire) {
  __shared__ AdamConsts cs;
  const size_t vec_end = (blk.n > 0 && (size_t)blk.end[0] <= n_grid) ? ((size_t)blk.end[0] & ~(size_t)3) : 0;
  // (skip_grid: rg.n = 0 -- nothing of block 0 is touched, not even its zero gradient)
  {   // lane b derives the constants of parameter block b (the lanes work side by side); all state words in ONE round trip
    const int b = threadIdx.x & (ADAM_MAX_BLOCKS - 1);
    const int found = si[2], t0 = si[4 + b], lab = counts ? counts[1] : 1;
    const float scale = sf[0], lrw = sf[1];
    if (threadIdx.x < ADAM_MAX_BLOCKS) {
      const bool has_sem = lab > 0;
      if (b == 0) { cs.skip = found ? 1.f : 0.f; cs.inv_scale = 1.0f / scale; }
      const float lr = lrw > 0.f ? lrw : h.lr;
      float ss = 0.f, ib = 1.f;
      if (b < blk.n && !found && adam_block_active(blk, b, has_sem)) {
        const int t = t0 + 1;
        const double bc1 = 1.0 - exp((double)t * h.log_beta1), bc2 = 1.0 - exp((double)t * h.log_beta2);   // 1 - beta^t
        ss = (float)((double)lr / bc1); ib = (float)(1.0 / sqrt(bc2));
      }
      cs.step_size[b] = ss; cs.inv_sqrt_bc2[b] = ib;
    }
  }
  __syncthreads();
  const bool skip = cs.skip != 0.f;
  const float inv_scale = cs.inv_scale;
  float r_step[ADAM_MAX_BLOCKS], r_isb[ADAM_MAX_BLOCKS];   // (registers: the loops below must not go back to LDS per element)
#pragma unroll
  for (int b = 0; b < ADAM_MAX_BLOCKS; ++b) { r_step[b] = b < blk.n ? cs.step_size[b] : 0.f; r_isb[b] = b < blk.n ? cs.inv_sqrt_bc2[b] : 1.f; }
  // Hash-grid block (14.2 M of the 14.3 M parameters; block 0 when it is the grid): four parameters per lane and trip, 16-byte
  // accesses.  In the step this kernel runs at the rate HBM sustains for an even read / write mix (16 B read + 18 B written per
  // parameter: 92-101 us = 4.8-5.3 TB/s; alone, with its 243 MB resident in the memory-side cache, 62-75 us:
  // scripts/dev/probe_adam.hip).  More loads in flight made it SLOWER in the step (next group requested before the current one is
  // written: 124 us; first group requested before the constants: 111 us): the limit is the DRAM bus turning around, not latency.
  {
    const float step_size = r_step[0], inv_sqrt_bc2 = r_isb[0];
    const bool idle = skip || step_size == 0.f;
    const float b1 = h.beta1, b2 = h.beta2, c1 = 1.f - h.beta1, c2 = 1.f - h.beta2;
    for (int k = 0; k < rg.n; ++k) {
    const size_t q_lo = (size_t)rg.lo[k] / 4, q_hi = min((size_t)rg.hi[k], vec_end) / 4;
    const long long mq0 = rg.mv[k] / 4 - (long long)q_lo;   // moment index of parameter group q: q + mq0
    for (size_t q = q_lo + blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < q_hi; q += (size_t)gridDim.x * blockDim.x) {
      const size_t mq = (size_t)((long long)q + mq0);
      // gwire (data parallelism, fp16 on the wire): the averaged table gradient is read as the halves the exchange left -- the same
      // fp32 values aln_grad_unpack_f16 would have written to g -- and there is no fp32 table gradient to read or to clear
      float4 g4;
      if (gwire) { if (idle) continue; const h16x4 w4 = *(const h16x4*)(gwire + 4 * q); g4 = make_float4((float)w4[0], (float)w4[1], (float)w4[2], (float)w4[3]); }
      else g4 = ((const float4*)g)[q];
      float4 p4, m4, v4;
      if (!idle) { p4 = ((const float4*)p)[q]; m4 = ((const float4*)m)[mq]; v4 = ((const float4*)v)[mq]; }
      if (!gwire) ((float4*)g)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idle) continue;
      float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
      h16x4 t4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gi = gg[k] * inv_scale;
        const float mi = b1 * mm[k] + c1 * gi;
        const float vi = b2 * vv[k] + c2 * gi * gi;
        mm[k] = mi; vv[k] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + h.eps;
        pp[k] -= step_size * (mi / denom);
        t4[k] = (h16)pp[k];
      }
      ((float4*)m)[mq] = make_float4(mm[0], mm[1], mm[2], mm[3]);
      ((float4*)v)[mq] = make_float4(vv[0], vv[1], vv[2], vv[3]);
      ((float4*)p)[q] = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *(h16x4*)(table16 + 4 * q) = t4;
    }
    }
  }
  // the rest (MLP weights, a ragged grid tail -- single-range launches only): one parameter per lane, block looked up per element
  const long long mi0 = rg.mlp_mv - (long long)n_grid;   // moment index of MLP parameter i: i + mi0 (0 for the replicated layout)
  for (size_t i = (blk.skip_grid || rg.mlp_mv != (long long)n_grid ? (size_t)blk.end[0] : vec_end) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_total; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    g[i] = 0.f;
    if (skip) continue;
    float step_size = r_step[0], inv_sqrt_bc2 = r_isb[0];
#pragma unroll
    for (int b = 1; b < ADAM_MAX_BLOCKS; ++b)
      if (b < blk.n && (long long)i >= blk.end[b - 1]) { step_size = r_step[b]; inv_sqrt_bc2 = r_isb[b]; }
    if (step_size == 0.f) continue;   // block without gradient this step (torch: grad is None)
    float pi = p[i];
    gi *= inv_scale;
    if (i >= n_grid) gi += h.wd_net * pi;
    const size_t im = (size_t)((long long)i + mi0);
    float mi = h.beta1 * m[im] + (1.f - h.beta1) * gi;
    float vi = h.beta2 * v[im] + (1.f - h.beta2) * gi * gi;
    m[im] = mi; v[im] = vi;
    float denom = sqrtf(vi) * inv_sqrt_bc2 + h.eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
    if (i < n_grid) table16[i] = (h16)pi;
  }
  // ---- arrival: the last block advances the optimizer / GradScaler state (torch.optim.Adam step counts, GradScaler.update)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(si + 15, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (int)gridDim.x - 1) {
      const float scale = sf[0];
      if (c_out) { c_out[0] = cs.skip; c_out[3] = cs.inv_scale; for (int b = 0; b < blk.n; ++b) { c_out[4 + 2 * b] = cs.step_size[b]; c_out[5 + 2 * b] = cs.inv_sqrt_bc2[b]; } }
      const bool has_sem = !counts || counts[1] > 0;
      if (skip) { sf[0] = scale * h.backoff; si[1] = 0; }
      else {
        si[0] += 1;
        for (int b = 0; b < blk.n; ++b) if (adam_block_active(blk, b, has_sem)) si[4 + b] += 1;
        int tr = si[1] + 1;
        if (tr >= h.growth_interval) { sf[0] = scale * h.growth; tr = 0; }
        si[1] = tr;
      }
      si[2] = 0;
      if (step_dev) *step_dev += 1u;   // the device-resident step number the RNG-consuming kernels of a captured step add to theirs
      __hip_atomic_store(si + 15, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// block_end[n_blocks] (host): exclusive end offsets of the parameter blocks in the flat buffer (last = n_total);
// block_kind[b]: 0 = always has a gradient, 1 = only with labelled rays (semantic_out), 2 = labelled rays or feature loss
// (semantic_features).  counts (device, from aln_loss_fwd_bwd) may be NULL: every block is then active.
// state_i: 16 int32 (si[15] = arrival ticket of the kernel: zero before the first launch, reset by the kernel itself).
static int adam_launch(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid,
                       int64_t n_total, int32_t* state_i, float* state_f, float* consts, float lr, float beta1,
                       float beta2, float eps, float wd_net, float growth, float backoff, int32_t growth_interval,
                       int32_t n_blocks, const int64_t* block_end, const int32_t* block_kind, int32_t feature_loss,
                       int32_t skip_grid, int32_t n_ranges, const int64_t* range_lo, const int64_t* range_hi,
                       const int32_t* counts, uint32_t* step_dev, const void* grid_wire_f16, void* stream) {
  ALN_REQUIRE(params && grads && m && v && state_i && state_f, "adam: NULL pointer");
  ALN_REQUIRE(!grid_wire_f16 || (n_ranges < 0 && !skip_grid && n_grid > 0 && n_grid % 4 == 0 && n_blocks > 0 && block_end[0] == n_grid &&
                                 ((uintptr_t)grid_wire_f16 & 7) == 0),
              "adam: the fp16 table gradient needs the replicated optimizer with the table as parameter block 0 (a multiple of 4 long)");
  ALN_REQUIRE(n_grid == 0 || table_f16, "adam: fp16 table shadow missing");
  ALN_REQUIRE(n_blocks >= 0 && n_blocks <= ADAM_MAX_BLOCKS && (n_blocks == 0 || (block_end && block_kind)), "adam: bad block table");
  ALN_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)table_f16 & 7) == 0,
              "adam: parameter / gradient / moment buffers must be 16-byte aligned (fp16 shadow: 8)");
  AdamHyper h{lr, beta1, beta2, eps, wd_net, growth, backoff, growth_interval, log((double)beta1), log((double)beta2)};
  AdamBlocks blk{};
  if (n_blocks == 0) { blk.n = 1; blk.end[0] = n_total; }
  else {
    blk.n = n_blocks;
    for (int b = 0; b < n_blocks; ++b) { blk.end[b] = block_end[b]; blk.needs_sem[b] = block_kind[b] == 1; blk.needs_sem_or_feat[b] = block_kind[b] == 2; }
  }
  blk.feat_on = feature_loss;
  ALN_REQUIRE(!skip_grid || (n_blocks > 0 && block_end[0] == n_grid), "adam: skip_grid needs the table as parameter block 0");
  blk.skip_grid = skip_grid ? 1 : 0;
  AdamRanges rg{};
  int64_t n_work = n_total - n_grid;
  if (n_ranges < 0) {   // the whole table (replicated optimizer), moments indexed like the parameters
    rg.mlp_mv = n_grid;
    if (!skip_grid && n_grid > 0) { rg.n = 1; rg.lo[0] = 0; rg.hi[0] = n_grid; rg.mv[0] = 0; n_work = n_total; }
  } else {
    ALN_REQUIRE(!skip_grid && n_ranges <= ADAM_MAX_RANGES && (n_ranges == 0 || (range_lo && range_hi)), "adam: bad range list (%d)", n_ranges);
    ALN_REQUIRE(n_grid % 4 == 0 && n_blocks > 0 && block_end[0] == n_grid, "adam: owned ranges need the table as parameter block 0, a multiple of 4 long");
    int64_t at = 0;
    for (int k = 0; k < n_ranges; ++k) {
      ALN_REQUIRE(0 <= range_lo[k] && range_lo[k] <= range_hi[k] && range_hi[k] <= n_grid && range_lo[k] % 4 == 0 && range_hi[k] % 4 == 0 &&
                  (k == 0 || range_lo[k] >= range_hi[k - 1]), "adam: owned range %d = [%lld, %lld)", k, (long long)range_lo[k], (long long)range_hi[k]);
      rg.lo[k] = range_lo[k]; rg.hi[k] = range_hi[k]; rg.mv[k] = at;
      at += range_hi[k] - range_lo[k];
    }
    rg.n = n_ranges; rg.mlp_mv = at; n_work += at;
  }
  // four parameters per thread at least: the arrival ticket at the end is ONE address, so the launch cannot be shorter than its
  // blocks' atomics one after the other (~20 ns each) -- 243 blocks for the 62 K MLP parameters of the single-GPU step took 10.7 us,
  // 2048 blocks for the 0.57 M of the LSeg heads 48 us (profiles/r05_lseg_leg_kernel_stats_rocprofv3.csv), for microseconds of work
  hipLaunchKernelGGL(k_adam, dim3(aln_grid_for(n_work > 0 ? (n_work + 3) / 4 : 1, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, (h16*)table_f16,
                     (size_t)n_grid, (size_t)n_total, state_i, state_f, consts, h, blk, rg, counts, step_dev, (const h16*)grid_wire_f16);
  ALN_CHECK_LAUNCH("adam");
  return 0;
}
extern "C" int aln_adam_step(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid,
                             int64_t n_total, int32_t* state_i, float* state_f, float* consts, float lr, float beta1,
                             float beta2, float eps, float wd_net, float growth, float backoff, int32_t growth_interval,
                             int32_t n_blocks, const int64_t* block_end, const int32_t* block_kind, int32_t feature_loss,
                             int32_t skip_grid, const int32_t* counts, uint32_t* step_dev, void* stream) {
  return adam_launch(params, grads, m, v, table_f16, n_grid, n_total, state_i, state_f, consts, lr, beta1, beta2, eps, wd_net, growth, backoff,
                     growth_interval, n_blocks, block_end, block_kind, feature_loss, skip_grid, -1, nullptr, nullptr, counts, step_dev, nullptr, stream);
}
// The same step with the TABLE's gradient read from the fp16 payload of the data-parallel exchange (grid_wire_f16[i] = averaged gradient of
// flat element i < n_grid, as aln_encode_bwd_binned_wire + the SUM all-reduce leave it; watched for non-finite halves by
// aln_grad_unpack_f16(grad = NULL) beforehand): bit for bit the step aln_grad_unpack_f16 + aln_adam_step take, without the fp32 copy of
// the table gradient (57 MB written, read and cleared per step).  grads[n_grid, n_total) (the MLP blocks) is read and cleared as usual.
extern "C" int aln_adam_step_wire(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid,
                                  int64_t n_total, int32_t* state_i, float* state_f, float* consts, float lr, float beta1,
                                  float beta2, float eps, float wd_net, float growth, float backoff, int32_t growth_interval,
                                  int32_t n_blocks, const int64_t* block_end, const int32_t* block_kind, int32_t feature_loss,
                                  const void* grid_wire_f16, const int32_t* counts, uint32_t* step_dev, void* stream) {
  ALN_REQUIRE(grid_wire_f16, "adam_step_wire: NULL wire buffer");
  return adam_launch(params, grads, m, v, table_f16, n_grid, n_total, state_i, state_f, consts, lr, beta1, beta2, eps, wd_net, growth, backoff,
                     growth_interval, n_blocks, block_end, block_kind, feature_loss, 0, -1, nullptr, nullptr, counts, step_dev, grid_wire_f16, stream);
}
// Sharded optimizer of the data-parallel engine (autolabel_amd/engine.py: shard_optimizer): the step touches only the slices
// [range_lo[k], range_hi[k]) of the hash table (ascending, disjoint, multiples of 4) and the whole MLP block.  `m`, `v` are
// COMPACT: the owned slices back to back, then the n_total - n_grid MLP moments.  Gradients outside the owned slices are neither
// read nor cleared (aln_grad_pack_f16 with clear_src does that when the gradient leaves for the reduce-scatter); the fp32 master
// and the fp16 table are updated inside the owned slices only (the table is then all-gathered).
extern "C" int aln_adam_step_ranges(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid,
                                    int64_t n_total, int32_t* state_i, float* state_f, float* consts, float lr, float beta1,
                                    float beta2, float eps, float wd_net, float growth, float backoff, int32_t growth_interval,
                                    int32_t n_blocks, const int64_t* block_end, const int32_t* block_kind, int32_t feature_loss,
                                    int32_t n_ranges, const int64_t* range_lo, const int64_t* range_hi, const int32_t* counts,
                                    uint32_t* step_dev, void* stream) {
  ALN_REQUIRE(n_ranges >= 0, "adam: negative range count");
  return adam_launch(params, grads, m, v, table_f16, n_grid, n_total, state_i, state_f, consts, lr, beta1, beta2, eps, wd_net, growth, backoff,
                     growth_interval, n_blocks, block_end, block_kind, feature_loss, 0, n_ranges, range_lo, range_hi, counts, step_dev, nullptr, stream);
}

// fp16 shadow of the grid table from the fp32 master (initialisation / checkpoint load)
__global__ void k_cast_f16(const float* __restrict__ src, h16* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (h16)src[i];
}
extern "C" int aln_cast_f16(const float* src, void* dst, int64_t n, void* stream) {
  ALN_REQUIRE(src && dst, "cast_f16: NULL pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_cast_f16, dim3(aln_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, src, (h16*)dst, (size_t)n);
  ALN_CHECK_LAUNCH("cast_f16");
  return 0;
}

__global__ void k_cast_f32(const h16* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
extern "C" int aln_cast_f32(const void* src_f16, float* dst, int64_t n, void* stream) {
  ALN_REQUIRE(src_f16 && dst, "cast_f32: NULL pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_cast_f32, dim3(aln_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const h16*)src_f16, dst, (size_t)n);
  ALN_CHECK_LAUNCH("cast_f32");
  return 0;
}

// ---- fp16 wire format of the data-parallel gradient exchange (autolabel_amd/parallel.py): the hash-grid gradient block is
// loss-scaled, i.e. already sized for fp16 (tcnn keeps these gradients in fp16 altogether), so it crosses xGMI as halves:
// out = fp16(g * mul) with mul = 1 / world (the SUM over the ranks is then the average and cannot overflow unless an input did),
// and back: g = fp32(in), raising found_inf for a non-finite element (the step is skipped and the scale backs off, exactly as
// for an fp16 overflow anywhere else).  8 elements per thread, 16-byte accesses on the fp16 side.
template <bool CLEAR>   // CLEAR: the source is zeroed behind the read and out[n, n_pad) is zero-filled (reduce-scatter staging)
__global__ void k_grad_pack_f16(float* __restrict__ g, size_t n, size_t n_pad, float mul, h16* __restrict__ out) {
  const size_t n8 = n / 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = *(const float4*)(g + 8 * i), b = *(const float4*)(g + 8 * i + 4);
    h16x8 v;
    v[0] = (h16)(a.x * mul); v[1] = (h16)(a.y * mul); v[2] = (h16)(a.z * mul); v[3] = (h16)(a.w * mul);
    v[4] = (h16)(b.x * mul); v[5] = (h16)(b.y * mul); v[6] = (h16)(b.z * mul); v[7] = (h16)(b.w * mul);
    *(h16x8*)(out + 8 * i) = v;
    if (CLEAR) { *(float4*)(g + 8 * i) = make_float4(0.f, 0.f, 0.f, 0.f); *(float4*)(g + 8 * i + 4) = make_float4(0.f, 0.f, 0.f, 0.f); }
  }
  if (blockIdx.x == 0) {
    for (size_t i = 8 * n8 + threadIdx.x; i < n; i += blockDim.x) { out[i] = (h16)(g[i] * mul); if (CLEAR) g[i] = 0.f; }
    if (CLEAR) for (size_t i = n + threadIdx.x; i < n_pad; i += blockDim.x) out[i] = (h16)0.f;
  }
}
__global__ void k_grad_unpack_f16(const h16* __restrict__ in, size_t n, float* __restrict__ g, int* __restrict__ found_inf) {
  const size_t n8 = n / 8;
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const h16x8 v = *(const h16x8*)(in + 8 * i);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] = (float)v[j]; bad |= !(fabsf(f[j]) <= 65504.f); }
    if (g) {   // (g == NULL: watch only -- the optimizer reads the halves itself, aln_adam_step_wire)
      *(float4*)(g + 8 * i) = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)(g + 8 * i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
  }
  if (blockIdx.x == 0) for (size_t i = 8 * n8 + threadIdx.x; i < n; i += blockDim.x) { const float f = (float)in[i]; bad |= !(fabsf(f) <= 65504.f); if (g) g[i] = f; }
  if (found_inf && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}
extern "C" int aln_grad_pack_f16(const float* grad, int64_t n, float mul, void* out_f16, void* stream) {
  ALN_REQUIRE(grad && out_f16 && n >= 0, "grad_pack_f16: bad arguments");
  ALN_REQUIRE(((uintptr_t)grad & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0, "grad_pack_f16: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_grad_pack_f16<false>, dim3(aln_grid_for(n / 8 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(grad), (size_t)n,
                     (size_t)n, mul, (h16*)out_f16);
  ALN_CHECK_LAUNCH("grad_pack_f16");
  return 0;
}
// Staging of a gradient bucket for a reduce-scatter: out[0, n) = fp16(grad * mul), out[n, n_pad) = 0 (the collective wants
// world equal shards), and grad[0, n) is cleared behind the read -- the scatter of the next step adds into it, and a rank with a
// sharded optimizer only ever clears the slice it owns.
extern "C" int aln_grad_pack_f16_clear(float* grad, int64_t n, int64_t n_pad, float mul, void* out_f16, void* stream) {
  ALN_REQUIRE(grad && out_f16 && n >= 0 && n_pad >= n, "grad_pack_f16_clear: bad arguments");
  ALN_REQUIRE(((uintptr_t)grad & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0, "grad_pack_f16_clear: buffers must be 16-byte aligned");
  if (n_pad == 0) return 0;
  hipLaunchKernelGGL(k_grad_pack_f16<true>, dim3(aln_grid_for(n / 8 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, grad, (size_t)n, (size_t)n_pad, mul,
                     (h16*)out_f16);
  ALN_CHECK_LAUNCH("grad_pack_f16_clear");
  return 0;
}
extern "C" int aln_grad_unpack_f16(const void* in_f16, int64_t n, float* grad, int32_t* found_inf, void* stream) {
  ALN_REQUIRE(in_f16 && n >= 0 && (grad || found_inf), "grad_unpack_f16: bad arguments");   // grad == NULL: only watch for non-finite halves
  ALN_REQUIRE(((uintptr_t)grad & 15) == 0 && ((uintptr_t)in_f16 & 15) == 0, "grad_unpack_f16: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_grad_unpack_f16, dim3(aln_grid_for(n / 8 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const h16*)in_f16, (size_t)n, grad,
                     found_inf);
  ALN_CHECK_LAUNCH("grad_unpack_f16");
  return 0;
}
