/* autolabel_hip.h -- C ABI of libautolabel_hip.so (MI355X / gfx950).
 *
 * The reference (ethz-asl/autolabel) has NO native code of its own: on this path it
 * calls two CUDA-only Python extensions, tinycudann and the torch-ngp fork.  Each entry
 * point below names the reference call site whose work it replaces (file:line relative
 * to the reference root) -- that is the interface a maintainer would bind (ctypes stub in
 * INTEGRATION.md).
 *
 * Conventions
 *  - plain pointers and sizes; every pointer is a DEVICE pointer unless marked host.
 *  - the caller owns all memory (PyTorch allocates it); kernels never allocate/sync.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*), asynchronous.
 *  - return 0 on success, negative on error; aln_last_error() gives the message
 *    (thread-local).
 *  - "rows" are samples in pass-major order: coarse rows ray*S1+i, then fine rows
 *    N*S1 + ray*S2 + j.  fp16 activations are row-major [rows, width].
 */
#ifndef AUTOLABEL_HIP_H
#define AUTOLABEL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ALN_MAX_LEVELS 16
#define ALN_ABI_VERSION 9

/* tcnn GridEncoding config, autolabel/models.py:38-48 */
typedef struct {
  int32_t n_levels, n_features, log2_hashmap_size, base_resolution;
  float per_level_scale;
  /* derived by aln_grid_desc_init */
  float scale[ALN_MAX_LEVELS];
  uint32_t res[ALN_MAX_LEVELS], size[ALN_MAX_LEVELS], offset[ALN_MAX_LEVELS], dense[ALN_MAX_LEVELS];
  uint32_t n_entries;
  int32_t pos_fma; /* 0: pos = x * scale + 0.5 in two fp32 roundings (this build's reproducible spec); 1: one fused multiply-add,
                      what tcnn's grid kernel compiles to -- for fields trained by the reference (checkpoint import) */
} AlnGridDesc;

/* input encoding: models.py:25-27 (freq), :51-59 (hg+freq), :142-143 (hg) */
typedef struct {
  int32_t n_freq;          /* 2 (hg+freq), 10 (freq), 0 (hg) */
  int32_t freq_normalized; /* 1: frequency encoding sees (x+b)/2b (FreqEncoder) */
  int32_t use_grid;
  int32_t enc_dim, enc_pad; /* 44 -> 48 */
  float bound;
  AlnGridDesc grid;
} AlnEncDesc;

/* One bias-free ReLU MLP (tcnn FullyFusedMLP / CutlassMLP; models.py:84-136).
 * Weights are fp16 in MFMA fragment order, produced by aln_mlp_repack from the fp32
 * row-major master copy  W_l[out_l][in_l]  (y = W x). */
typedef struct {
  int32_t in_pad, hidden, out_pad, n_hidden; /* n_hidden in {1,2}; hidden in {64,128} */
  const void* wf;   /* forward fragments  */
  const void* wb;   /* backward (transposed) fragments */
  const void* wr;   /* row-major fp16 copy, pitch in+8 (recompute backward); may be NULL */
  void* dw_ws;      /* scratch for the recompute backward's per-block weight-gradient partial sums, >= aln_mlp_dw_ws_bytes();
                       required when aln_mlp_bwd / aln_sem_heads_bwd are asked for dW on the recompute path (slabs are reduced in
                       a fixed order: bit-reproducible weight gradients, no atomics) */
  int64_t dw_ws_bytes;
  int32_t defer_dw_reduce; /* 1: the recompute backward leaves its slabs in dw_ws; the caller folds them into dW later with
                              aln_mlp_dw_reduce_all (all heads of a training step in one launch) */
  int32_t x_tiled;  /* 1: the input rows of aln_density_fwd / aln_mlp_fwd / aln_mlp_bwd are in the 32-row TILED layout that
                       aln_encode_fwd_phased(planes_ws = NULL) writes -- tile t = rows [32 t, 32 t + 32), inside it the in_pad / 8
                       16-byte pieces of a row piece-major: piece p of row r at halves 32 in_pad t + 256 p + 8 (r % 32) -- i.e. exactly the
                       per-lane pieces the 128-wide kernels fetch (mlp_fwd128.hip, mlp_bwd128.hip).  Only those kernels read it
                       (aln_mlp_supports_tiled); the buffer must hold whole tiles.
                       2 (ABI 8): PAIR PLANES -- what aln_encode_fwd_planes writes: x points at in_pad / 2 planes of 4-byte words, plane q
                       holding features (2 q, 2 q + 1) of every row, word r of plane q at x + 4 (q x_pitch + r) bytes.  The same kernels
                       read it (their loaders fetch whole 16-byte pieces of a plane); the forward needs rows % 32 == 0. */
  int64_t x_pitch;  /* x_tiled = 2: words between two planes (>= the rows of the widest launch; a multiple of 4) */
} AlnMlpDesc;

const char* aln_last_error(void);
int aln_abi_version(void);
int aln_grid_desc_init(AlnGridDesc* g); /* host */
int64_t aln_mlp_dw_ws_bytes(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden); /* host: size of AlnMlpDesc.dw_ws */

/* ---- ray generation: autolabel/dataset.py:17-37 (_compute_direction), :182-242
 * (_next_train), :244-266 (_get_test), on device-resident frames. */
typedef struct {
  const float* images;      /* [F, H*W, 3] f32 */
  const uint16_t* depths;   /* [F, H*W] u16 millimetres */
  const uint8_t* semantics; /* [F, H*W] u8, 0 = unlabeled */
  const void* features;     /* [F, Hf*Wf, Cf] f16 or NULL */
  const float* rotations;   /* [F,3,3] f32 (R_WC, ngp axes) */
  const float* origins;     /* [F,3] */
  const int32_t* pixel_indices; /* [n_pix] valid pixels (dataset.py:295-311) */
  int32_t n_frames, w, h, n_pix, feat_w, feat_h, feat_c;
  double fx, fy, cx, cy;
  /* optional class index (IndexSampler, dataset.py:80-151): for class k, cls_offsets[k*(n_frames+1) + f .. f+1] delimit the
   * pixels of that class in frame f inside cls_pixels; a labelled chunk (probability sem_ratio, dataset.py:207-211) draws a
   * class uniformly, a frame proportionally to its pixel count of that class, then pixels of that class in that frame. */
  const int32_t* cls_offsets; /* [n_classes, n_frames+1] */
  const int32_t* cls_pixels;
  int32_t n_classes;
  float sem_ratio;            /* 0.5 in the reference; 0 disables */
} AlnFrames;

typedef struct {
  float* rays_o;  /* [B,3] */
  float* rays_d;  /* [B,3] */
  float* norms;   /* [B]   */
  float* pixels;  /* [B,3] */
  float* depth;   /* [B] metres */
  int32_t* semantic; /* [B], -1 = unlabeled */
  float* features;   /* [B,Cf] or NULL */
} AlnBatch;

/* chunk_frames[B/chunk] (device) picks the frame per 512-ray chunk; if NULL frames are
 * drawn with the counter RNG.  pixel ids / jitter: counter RNG(seed, step), or explicit
 * arrays for parity tests (ray_idx [B] i32, jitter [B,2] f32). */
int aln_raygen_train(const AlnFrames* fr, const AlnBatch* out, int32_t B, int32_t chunk,
                     int32_t frame_lo, int32_t frame_hi, uint32_t seed, uint32_t step,
                     const int32_t* chunk_frames, const int32_t* ray_idx, const float* jitter,
                     const uint32_t* step_dev /* device word added to step, or NULL: lets a captured hipGraph replay with a
                                                 fresh step number */, void* stream);
int aln_raygen_frame(const AlnFrames* fr, const AlnBatch* out, int32_t frame, void* stream);
/* bare _compute_direction on an index list (fixture F1) */
int aln_compute_direction(const float* R_WC, const int64_t* idx, int32_t n, int32_t w, double fx, double fy,
                          double cx, double cy, const float* jitter, float* dirs, float* norms, void* stream);

/* ---- sampling: torch-ngp NeRFRenderer.run reached from autolabel/trainer.py:64-70 */
/* near/far of each ray against [-bound,bound]^3 (raymarching.near_far_from_aabb; a miss gives near = far = min_near) */
int aln_ray_aabb(const float* rays_o, const float* rays_d, int32_t N, float bound, float min_near, float* nears,
                 float* fars, void* stream);
int aln_sample_coarse(const float* rays_o, const float* rays_d, int32_t N, int32_t S1, float bound, float min_near,
                      int32_t perturb, uint32_t seed, uint32_t step, const float* noise /*[N,S1] or NULL*/,
                      float* nears, float* fars, float* z /*[N,S1]*/, const uint32_t* step_dev /*see aln_raygen_train*/,
                      void* stream);
int aln_sample_fine(const float* z_coarse, const float* sigma_coarse, const float* nears, const float* fars, int32_t N,
                    int32_t S1, int32_t S2, float density_scale, int32_t perturb, uint32_t seed, uint32_t step,
                    const float* u /*[N,S2] or NULL*/, float* z_fine /*[N,S2] sorted*/, const uint32_t* step_dev,
                    void* stream);

/* ---- encoding: tcnn Frequency + GridEncoding, autolabel/models.py:51-59 */
int aln_encode_fwd(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d,
                   const float* z /*[rows]*/, const float* xyz /*[rows,3] or NULL*/, int32_t rows,
                   int32_t rays_stride /*samples per ray for this pass*/, void* enc_out /*[rows,enc_pad] f16*/,
                   void* stream);
/* the same result for large row counts, level-phased so that the tables in flight stay L2-resident (encode.hip); planes_ws is
 * caller-owned scratch of aln_encode_fwd_ws_bytes(e, rows) bytes.  planes_ws = NULL (round 5): no planes and no assembly pass -- every
 * level's wave writes its features straight into enc_out in the TILED layout of AlnMlpDesc.x_tiled (the four levels of a 16-byte
 * piece are written by blocks of the same XCD a few microseconds apart and merge in its L2); enc_out must hold ceil(rows / 32) whole
 * tiles and enc_pad must be 32 or 48 */
int64_t aln_encode_fwd_ws_bytes(const AlnEncDesc* e, int32_t rows);
int aln_encode_fwd_phased(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d,
                          const float* z, const float* xyz, int32_t rows, int32_t rays_stride, void* planes_ws, void* enc_out,
                          void* stream);
/* The level-phased gather writing the density head's input as PAIR PLANES (AlnMlpDesc.x_tiled = 2, round 6): plane q of planes_out =
 * the fp16 pair (2 q, 2 q + 1) of every row's encoding -- enc_pad / 2 planes in row order of the features (frequency pairs, one plane
 * per level, the ones padding), each wave's 64 rows of a plane one coalesced 256-byte store, plane_pitch words between two planes.
 * No assembly pass and no second buffer: the 128-wide forward and backward kernels read the planes themselves.  A launch over rows
 * [a, a + rows) of a larger batch passes planes_out + a words (both sampling passes of a step share one set of planes). */
int aln_encode_fwd_planes(const AlnEncDesc* e, const void* table_f16, const float* rays_o, const float* rays_d, const float* z,
                          const float* xyz, int32_t rows, int32_t rays_stride, void* planes_out, int64_t plane_pitch, void* stream);
/* encoding of one jittered point per occupancy-grid cell, cells [cell0, cell0 + rows) of a G^3 grid over [-bound,bound]^3 (the
 * positions of aln_grid_points, generated inside the kernel): input of the density head for the density-grid refresh,
 * NeRFRenderer.update_extra_state at autolabel/trainer.py:34-36.  planes_ws != NULL selects the level-phased kernels. */
int aln_encode_fwd_cells(const AlnEncDesc* e, const void* table_f16, int32_t G, uint32_t seed, uint32_t step,
                         const uint32_t* step_dev, int32_t cell0, int32_t rows, void* planes_ws, void* enc_out, void* stream);
/* backward: grad_table [n_entries * F] f32 += scatter of w_corner * d_enc, WITHOUT global atomics (encode.hip, "binned
 * backward"): phase 1 counting-sorts the run-deduped (index, value) records of every 512-row tile by table slice (1/64 of a
 * level) and streams them to `ws`; phase 2 accumulates each slice in LDS in 64-bit fixed point (exact, order-independent: the
 * result is bit-reproducible) and adds it to grad_table.  [level_lo, level_hi): a data-parallel caller launches the levels in
 * groups and all-reduces the finished part of the table while the next group is still being scattered.  Rows [0, rows_pass1) are rays_stride1 samples per ray, the rest rays_stride2 (the coarse and
 * the importance pass of autolabel/trainer.py:64-70 in one launch).  ws = aln_encode_bwd_binned_ws_bytes(e, rows) bytes of
 * caller-owned scratch.  *found_inf is set when a gradient entry is not finite (record values travel as fp16). */
/* Optional optimizer for the table, applied by phase 2 itself (single-GPU training: the exact gradient sums of a slice sit in LDS,
 * so the block that owns the slice takes the Adam step for its entries and the gradient never travels through HBM).  Same
 * arithmetic as aln_adam_step, whose state words it reads (state_i[2] found_inf, state_i[4] steps of block 0, state_f[0] loss
 * scale, state_f[1] learning rate); aln_adam_step(..., skip_grid = 1) afterwards updates the MLP blocks and advances the state. */
typedef struct {
  float* params; float* m; float* v; void* table_f16;   /* the grid part of the flat buffers (element 0 = entry 0 of level 0) */
  const int32_t* state_i; const float* state_f;
  float lr, beta1, beta2, eps;
} AlnAdamFuse;
int64_t aln_encode_bwd_binned_ws_bytes(const AlnEncDesc* e, int32_t rows);
int32_t aln_encode_bwd_binned_tile_rows(void);   /* sample rows per phase-1 tile (layout of ws: per (level, tile) a pool chunk with room for 8 x tile PAIR records of 12 bytes --
                                                  * slot0 | slot1 << 13, fp16x2 value 0, fp16x2 value 1 -- then the [level][slice][tile] descriptors) */
int aln_encode_bwd_binned(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z, const float* xyz,
                          int32_t rows, int32_t rows_pass1, int32_t rays_stride1, int32_t rays_stride2,
                          const uint16_t* depth_order /* optional [rays, stride1 + stride2] */, const void* d_enc,
                          float* grad_table, void* ws, int32_t level_lo, int32_t level_hi, int32_t* found_inf,
                          const AlnAdamFuse* adam /* NULL: gradients are added to grad_table */, void* stream);
/* Data parallelism with fp16 on the wire (no counterpart in the reference, which has no multi-GPU path: scripts/train.py:83):
 * the same scatter, but the gradient of levels [level_lo, level_hi) leaves as the PAYLOAD of the exchange --
 * wire_f16[2 * (offset[l] + entry) + feature] = fp16(float(sum) * wire_mul) for every entry of those levels, zeros included -- bit for bit what
 * aln_grad_pack_f16(mul = wire_mul) makes of the table aln_encode_bwd_binned adds into when it starts from zeros.  No fp32 gradient table
 * is read or written (ABI 7). */
int aln_encode_bwd_binned_wire(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z, const float* xyz,
                               int32_t rows, int32_t rows_pass1, int32_t rays_stride1, int32_t rays_stride2,
                               const uint16_t* depth_order, const void* d_enc, void* ws, int32_t level_lo, int32_t level_hi,
                               int32_t* found_inf, void* wire_f16 /* fp16 [2 * table entries] */, float wire_mul /* 1 / world */, void* stream);
/* The two phases of the scatter on their own, for the overlapped exchange: phases = 1 writes the records of levels [level_lo, level_hi)
 * into ws (one launch for all levels), phases = 2 accumulates a level range from ws into grad_table (added) or, with wire_f16 != NULL, into
 * the fp16 payload; phases = 3 = aln_encode_bwd_binned / _wire.  Bit-identical results however the levels are grouped (ABI 7). */
int aln_encode_bwd_binned_phase(const AlnEncDesc* e, const float* rays_o, const float* rays_d, const float* z, const float* xyz,
                                int32_t rows, int32_t rows_pass1, int32_t rays_stride1, int32_t rays_stride2,
                                const uint16_t* depth_order, const void* d_enc, float* grad_table, void* ws, int32_t level_lo,
                                int32_t level_hi, int32_t* found_inf, void* wire_f16, float wire_mul, int32_t phases, void* stream);

/* ---- MLPs: tcnn Network{FullyFusedMLP,CutlassMLP}, autolabel/models.py:84-136 */
int aln_mlp_repack(const float* w_master, int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden,
                   void* wf, void* wb, void* wr /*optional*/, void* stream);
/* the same for n_heads descriptors (wf / wb / wr taken from each AlnMlpDesc) in ONE launch: the per-step refresh of the
 * fp16 copies after the optimizer (scripts/train.py:50-63 keeps fp32 masters; tcnn re-casts them every step) */
int aln_mlp_repack_all(int32_t n_heads, const float* const* w_master, const AlnMlpDesc* const* descs, void* stream);
int64_t aln_mlp_frag_halves(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden, int32_t backward);
int64_t aln_mlp_rowmajor_halves(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden);
/* h1/h2: [rows,hidden] f16 saved post-ReLU activations (NULL at inference).  rows_dev (optional, device int32)
 * clamps the row count on the device (compacted live samples). */
int aln_mlp_fwd(const AlnMlpDesc* m, const void* x, int32_t rows, const int32_t* rows_dev, void* h1, void* h2,
                void* out, void* stream);
/* ALNetwork.density (autolabel/models.py:175-188) in one launch: out [rows,16] = sigma_net(x) and sigma[row] = trunc_exp(out[row][0])
 * (fp32, from the fp16 output: the epilogue form of aln_sigma_act) */
int aln_density_fwd(const AlnMlpDesc* m, const void* x, int32_t rows, void* h1, void* h2, void* out, float* sigma, void* stream);
/* h1 == NULL selects the recompute backward: hidden activations are rebuilt from x inside the kernel (needs m->wr).
 * d_in (optional) [rows,in_pad] f16; dW (optional) += weight gradients in the fp32 master layout; dA1/dA2 are scratch
 * [rows,hidden] f16 (only touched by the unfused fallback); found_inf is OR-ed when an fp16 gradient overflows. */
/* 1 if the recompute backward exists for this shape (otherwise pass saved activations to aln_mlp_bwd) */
/* 1 when this head's forward / backward kernels can read AlnMlpDesc.x_tiled input rows (128 wide, two hidden layers, 16 outputs, 32 or
 * 48 inputs) */
int aln_mlp_supports_tiled(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden);
int aln_mlp_has_recompute(int32_t in_pad, int32_t hidden, int32_t out_pad, int32_t n_hidden);
int aln_mlp_bwd(const AlnMlpDesc* m, const void* x, const void* h1, const void* h2, const void* d_out, int32_t rows,
                const int32_t* rows_dev, void* dA1, void* dA2, void* d_in, float* dW, int32_t* found_inf, void* stream);
/* aln_mlp_bwd (recompute) of the DENSITY head with its dL/dout rows assembled inside the kernel's loader (ABI 8): row r =
 * [ d_h0[r] | d_semf_in[r][0 .. G) + d_color_in[cidx_row[r]][16 .. 16 + G) ] (the colour term where cidx_row[r] >= 0), fp32 sums rounded
 * once to fp16 -- what aln_assemble_grads(d_semo_in = NULL) writes into d_sigma_out (autolabel/models.py:175-188: the density head's
 * output row is [sigma logit | geo_feat], and geo_feat feeds the colour and the semantic heads); that pass and its buffer are not
 * needed.  d_semf_in [rows, 16] f16, d_color_in [live rows, 32] f16, 128-wide head with 48 inputs only (-3 otherwise). */
int aln_mlp_bwd_dso(const AlnMlpDesc* m, const void* x, const float* d_h0, const void* d_semf_in, const void* d_color_in,
                    const int32_t* cidx_row, int32_t G, int32_t rows, void* d_in, float* dW, int32_t* found_inf, void* stream);
/* deferred weight-gradient reduction (AlnMlpDesc.defer_dw_reduce): dW[k] += sum of the slabs head k's last recompute backward
 * over rows[k] rows left in its dw_ws, all heads in ONE launch; aln_mlp_bwd_blocks = the number of slabs (the backward's grid) */
int32_t aln_mlp_bwd_blocks(const AlnMlpDesc* m, int32_t rows);
int aln_mlp_dw_reduce_all(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, void* stream);
/* the same with the slab count of each head given explicitly (slabs[k] > 0: what aln_sem_heads_bwd_slabs returned for the fused
 * semantic pair; 0: aln_mlp_bwd_blocks(descs[k], rows[k])) */
int aln_mlp_dw_reduce_slabs(int32_t n_heads, const AlnMlpDesc* const* descs, float* const* dW, const int32_t* rows, const int32_t* slabs,
                            void* stream);

/* Both semantic heads (models.py:248-256) with their inputs / output gradients built on the fly from sigma_out, f, the
 * compositing weights and the per-ray output gradients (no [rows,80] / [rows,64] intermediates in HBM).
 * d_semo_in [rows, D + 16]: columns [0, D) = dL/df of the semantic_out branch (the ReLU mask of cat[relu(f), geo_feat] applied),
 * columns [D, D + 16) = its dL/d(geo_feat, 1). */
int aln_sem_heads_fwd(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, int32_t rows, int32_t D, int32_t G,
                      void* feat, void* logits, void* stream);
/* the training step's forward: no rows leave the kernel, only tile_sums[ceil(rows / 32)][96] = per 32-row tile the sums of
 * w_row * f (64 columns) and w_row * logits (out_pad <= 32 columns) -- what models.py:195-203's weighted sums need of them.  Same
 * shapes as the one-kernel backward (aln_sem_heads_bwd_slabs > 0); a tile must not straddle two rays (S1, S2 multiples of 32). */
int aln_sem_heads_fwd_sums(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, int32_t rows, int32_t D, int32_t G,
                           const float* w_row, float* tile_sums, void* stream);
int aln_sem_heads_bwd(const AlnMlpDesc* semf, const AlnMlpDesc* semo, const void* sigma_out, const void* feat, const float* w_row,
                      const float* g_sem, const float* g_feat, int32_t N, int32_t S1, int32_t S2, int32_t C, int32_t rows,
                      int32_t D, int32_t G, void* d_semo_in, void* d_semf_in, float* dW_semf, float* dW_semo,
                      int32_t fold_geo /* 1 = d_semf_in rows also take the geo_feat columns of d_semo_in, so that
                                          aln_assemble_grads needs d_semf_in only (d_semo_in = NULL there) */,
                      float* dots_row /* optional [rows], one-kernel form only: <logits_s, g_sem[ray]> + <f_s, g_feat[ray]> per sample row,
                                         the semantic outputs' share of dL/dw_s (S1, S2 multiples of 32; g in fp16 like every gradient of the chain) -- aln_composite_bwd(dots_row) then needs neither f nor the
                                         logits, and the forward need not store them (aln_sem_heads_fwd_sums) */,
                      int32_t* found_inf, void* stream);
/* slabs per head that aln_sem_heads_bwd(..., fold_geo = 1, both dW) leaves in semf->dw_ws / semo->dw_ws when the pair runs as ONE
 * kernel (both heads 64 wide, D = 64, <= 32 padded classes: models.py:248-256 at the reference's sizes); 0 = two launches, the
 * per-head aln_mlp_bwd_blocks apply.  In the fused form d_semo_in is not written (dL/df never leaves the CU) and `feat` is not read. */
int32_t aln_sem_heads_bwd_slabs(const AlnMlpDesc* semf, const AlnMlpDesc* semo, int32_t rows, int32_t D, int32_t G);
/* ---- head plumbing: autolabel/models.py:175-188 (sigma = trunc_exp(h0), geo_feat = h[1:]), :190-220 (boolean-mask
 * gather + SH(dir) ++ geo_feat), :248-256 (cat[relu(f), geo_feat]) and the matching gradient assembly */
int aln_sigma_act(const void* sigma_out /*[rows,16] f16*/, int32_t rows, float* sigma, void* stream);
/* live rows (w_row > thresh: the renderer's `weights > 1e-4` mask, autolabel/models.py:199-203) compacted in row order (a pure
 * function of w_row: deterministic); chunk_ws = aln_compact_live_ws_ints(rows) int32 of caller-owned scratch */
int32_t aln_compact_live_ws_ints(int32_t rows);
int aln_compact_live(const float* w_row, int32_t rows, float thresh, int32_t* n_live, int32_t* live_idx, int32_t* cidx_row,
                     int32_t* chunk_ws, void* stream);
/* aln_compact_live + aln_build_color_in in the compaction's second pass (round 6, ABI 9): color_in[cidx_row[r]] is written for
 * every live row r where its place becomes known -- the same rows bit for bit, one launch and one read of live_idx less */
int aln_compact_live_color_in(const float* w_row, int32_t rows, float thresh, int32_t* n_live, int32_t* live_idx, int32_t* cidx_row,
                              int32_t* chunk_ws, const float* rays_d, const float* dirs, int32_t N, int32_t S1, int32_t S2,
                              const void* sigma_out, int32_t G, int32_t in_pad, void* color_in, void* stream);
/* tcnn SphericalHarmonics(degree 4) of the remapped direction (autolabel/models.py:97-103,205-207): out[r, 0:16] f16 */
int aln_sh4(const float* dirs /*[rows,3]*/, int32_t rows, int32_t out_pitch /*halves, >= 16*/, void* out, void* stream);
/* color_net forward with its input rows built inside the kernel from live_idx / directions / sigma_out (inference path:
 * no color_in tensor; dirs = one direction per row [rows,3], else rays_d [N,3] indexed through the row's ray) */
int aln_color_fwd(const AlnMlpDesc* color, const int32_t* live_idx, const int32_t* n_live, int32_t max_rows,
                  const float* rays_d, const float* dirs, int32_t N, int32_t S1, int32_t S2, const void* sigma_out, int32_t G,
                  void* color_out, void* stream);
int aln_build_color_in(const int32_t* live_idx, const int32_t* n_live, int32_t max_rows, const float* rays_d,
                       const float* dirs, int32_t N, int32_t S1, int32_t S2, const void* sigma_out, int32_t G, int32_t in_pad,
                       void* color_in, void* stream);
int aln_build_sem_in(const void* sigma_out, const void* f, int32_t rows, int32_t D, int32_t G, int32_t semf_in_pad,
                     int32_t semo_in_pad, void* semf_in, void* semo_in, void* stream);
int aln_assemble_grads(const float* d_h0, const void* d_semf_in, int32_t semf_in_pad, const void* d_semo_in,
                       int32_t semo_in_pad, int32_t D, const void* d_color_in, int32_t color_in_pad, const int32_t* cidx_row,
                       int32_t rows, int32_t G, void* d_sigma_out, int32_t* found_inf, void* stream);
int aln_assemble_dsemf_out(void* d_feat, const void* f, const void* d_semo_in, int32_t rows, int32_t D, int32_t semo_in_pad,
                           int32_t* found_inf, void* stream);

/* open-vocabulary prompt comparison (OpenVocabEvaluator._predict_semantic, autolabel/evaluation.py:295-327, 400-445):
 * out[r] = argmax_c <features[r, :], text[c, :]> in fp32 (first maximum; an all-zero feature row -> 0, as the reference's NaN row) */
int aln_similarity_argmax(const float* features /*[n,D]*/, int32_t n, int32_t D, const float* text /*[C,D]*/, int32_t C,
                          int64_t* out /*[n]*/, void* stream);

/* ---- wide semantic heads (wide.hip): semantic_features / semantic_out at LSeg width (hidden_dim_semantic = 512,
 * autolabel/models.py:117-136, scripts/ros/node.py:166-176), one hand-written MFMA GEMM launch per layer.
 * A operand = [a1 (K1 columns, optionally through ReLU) | geo block (16 columns [geo_feat, 1..] built from the density head's
 * output rows sigma_out[M,16]) when geo != NULL]. */
/* Y[M,N] = epi(A W[N,K]^T): epi = (* (mask > 0)) (+ add) (ReLU), fp16 out; found_inf raised on a non-finite output */
int aln_wide_nt(const void* a1, int32_t lda1, int32_t K1, int32_t relu1, const void* geo, int32_t G, int32_t M, int32_t N,
                const void* w, int32_t ldw, void* y, int32_t ldy, int32_t relu, const void* mask, int32_t ldm, const void* add,
                int32_t lda, int32_t* found_inf, void* stream);
/* dW[N,K] (fp32, row-major, leading dimension lddw) += G[M,N]^T A[M,K]: row ranges of M go to blocks that write partial sums into
 * `ws` (aln_wide_tn_ws_bytes(M, N, K) bytes, K = K1 + 16 with geo), added to dW in a fixed order -- no atomics, bit-reproducible */
int64_t aln_wide_tn_ws_bytes(int32_t M, int32_t N, int32_t K);
int aln_wide_tn(const void* g, int32_t ldg, const void* a1, int32_t lda1, int32_t K1, int32_t relu1, const void* geo, int32_t G,
                int32_t M, int32_t N, float* dw, int32_t lddw, void* ws, void* stream);
int aln_transpose_f16(const void* src /*[R,C]*/, int32_t R, int32_t C, void* dst /*[C,R]*/, void* stream);
/* "Generated" first hidden layer (round 5): h1 = relu([geo_feat, 1] W0[K,16]^T) of semantic_features (autolabel/models.py:117-125 at LSeg
 * width) is one matrix instruction per 32 x 32 block, so it is recomputed wherever it is needed instead of being stored (1 GB at 2^20
 * rows x 512) and read back three times:
 *   aln_wide_nt_gen      Y[M,N] = epi(h1 W[N,K]^T): layers 1 + 2 in one launch.  w_perm = W with the columns of every group of 16 in
 *                        the order 0-3, 8-11, 4-7, 12-15 (the generated registers are the B operands in that contraction order)
 *   aln_wide_nt_maskgen  Y[M,N] = (A[M,K] W[N,K]^T) * (h1 > 0): the data gradient flowing into the generated layer (h1: [M,N], W0: [N,16])
 *   aln_wide_tn_gen      dW[N,K] += G[M,N]^T h1[M,K]: the weight gradient of the layer behind it (ws: aln_wide_tn_ws_bytes(M, N, K)) */
int aln_wide_nt_gen(const void* geo /*sigma_out [M,16]*/, int32_t G, const void* w0 /*[K,16] f16*/, int32_t M, int32_t N, int32_t K,
                    const void* w_perm, int32_t ldw, void* y, int32_t ldy, int32_t relu, const float* w_row /*[M] or NULL*/,
                    float* tile_sums /*[ceil(M/32), N] or NULL: sum over each 32-row tile of w_row[row] * Y[row][:]*/, int32_t* found_inf,
                    void* stream);
/* aln_composite_out for a per-sample activation whose producer left its per-tile weighted sums (aln_wide_nt_gen(tile_sums)): the image
 * as usual, features[ray] = the ray's tiles added up in a fixed order; no [M, D] rows are read */
int aln_composite_out_featsums(const float* w_row, const int32_t* cidx_row, const void* color_out, const float* wsum, int32_t N, int32_t S1,
                               int32_t S2, int32_t D, float bg, float* image, float* features, const float* feat_sums, void* stream);
int aln_wide_nt_maskgen(const void* a1, int32_t lda1, int32_t M, int32_t N, int32_t K, const void* w, int32_t ldw, void* y, int32_t ldy,
                        const void* geo, int32_t G, const void* w0 /*[N,16]*/, int32_t* found_inf, void* stream);
int aln_wide_tn_gen(const void* g, int32_t ldg, const void* geo, int32_t G, const void* w0 /*[K,16]*/, int32_t M, int32_t N, int32_t K,
                    float* dw, int32_t lddw, void* ws, void* stream);
/* both consumers of the data gradient g = dL/dh1 [M,N] (N <= 512) that flows into the generated layer, in ONE pass over it:
 * d_fin[M,16] (fp16) = g W0 with w0t = W0^T [16,N], and dW0[N,16] += g^T [geo_feat, 1] (ws: aln_wide_tn_din_ws_bytes(M, N)) */
int64_t aln_wide_tn_din_ws_bytes(int32_t M, int32_t N);
int aln_wide_tn_din(const void* g, int32_t ldg, const void* geo, int32_t G, const void* w0t, int32_t ldw0t, int32_t M, int32_t N, float* dw,
                    int32_t lddw, void* ws, void* d_fin, int32_t* found_inf, void* stream);

/* ---- occupancy-grid marching (march.hip): the cuda_ray hooks of autolabel/trainer.py:21-23,34-36,176 and
 * NeRFRenderer.mark_untrained_grid / update_extra_state of the torch-ngp fork (dead in the reference: model_utils.py:72).
 * One-level G^3 density grid over [-bound,bound]^3 (-1 = never seen), one bit per cell. */
/* every ray gets exactly S sample rows z[N,S] + step lengths delta[N,S] inside occupied cells (evenly subsampled when more than
 * S of the max_steps uniform steps are occupied, padded with delta = 0 otherwise); counts[N] (optional) = occupied steps */
int aln_march_rays(const float* rays_o, const float* rays_d, int32_t N, int32_t S, float bound, float min_near,
                   const uint32_t* bitfield, int32_t G, int32_t max_steps, int32_t perturb, uint32_t seed, uint32_t step,
                   const uint32_t* step_dev, const float* noise /*[N] or NULL*/, float* nears, float* fars, float* z, float* delta,
                   int32_t* counts, void* stream);
/* one jittered point per cell, xyz[G^3,3] (input of the density head for the grid update) */
int aln_grid_points(int32_t G, float bound, uint32_t seed, uint32_t step, const uint32_t* step_dev /*see aln_raygen_train*/,
                    const float* noise /*[G^3,3] or NULL*/, float* xyz, void* stream);
/* grid = max(grid * decay, sigma * density_scale) on cells >= 0 (sigma NULL: statistics only), then
 * bit = grid > min(mean over cells >= 0, thresh); stats = 16 bytes of scratch (the mean is accumulated in integers: the
 * bitfield does not depend on the order the blocks arrive in); n_set (optional) = number of set bits */
int aln_grid_update(float* grid, const float* sigma, int32_t G, float decay, float density_scale, float thresh, void* stats,
                    uint32_t* bitfield, int32_t* n_set, void* stream);
/* n_set = population count of a bitfield (a checkpoint's density_bitfield is kept as stored, autolabel/model_utils.py:9-18) */
int aln_bitfield_count(const uint32_t* bitfield, int64_t n_words, int32_t* n_set, void* stream);
/* cells no camera sees become -1; T_CW [n_poses,4,4] row-major world -> OpenCV camera, pinhole fx fy cx cy, image w x h */
int aln_mark_untrained_grid(float* grid, int32_t G, float bound, const float* T_CW, int32_t n_poses, float fx, float fy,
                            float cx, float cy, float w, float h, float z_near, int32_t sub, void* stream);

/* ---- compositing: alpha / cumprod weights and weighted sums of NeRFRenderer.run (fork), outputs image, depth
 * (metric z-depth = sum w t / direction_norm), semantic, semantic_features, depth_variance, coordinates_map;
 * call sites autolabel/trainer.py:64-70, scripts/export.py:83-89, scripts/language/pointcloud.py:58-68 */
int aln_composite_fwd(const float* rays_o, const float* rays_d, const float* norms, const float* nears, const float* fars,
                      const float* z, const float* sigma, int32_t N, int32_t S1, int32_t S2, float bound, float density_scale,
                      uint16_t* perm, float* w_row, float* T_row, float* delta_row, float* wsum, float* depth, float* depth_var,
                      float* coords, const float* delta_in /*[N,S1] explicit step lengths (marching, S2 == 0) or NULL*/,
                      void* stream);
int aln_composite_out(const float* w_row, const int32_t* cidx_row, const void* color_out, const void* logits, const void* feat,
                      const float* wsum, int32_t N, int32_t S1, int32_t S2, int32_t C, int32_t Cpad, int32_t D, float bg,
                      float* image, float* semantic, float* features,
                      const float* tile_sums /* optional [rows / 32][96]: the per-tile weighted sums of aln_sem_heads_fwd_sums in place of
                                                the logits / feat rows (both NULL then; D = 64, C <= 32, S1 and S2 multiples of 32) */,
                      void* stream);
int aln_composite_bwd(const float* norms, const float* z, const float* sigma, const uint16_t* perm, const float* w_row,
                      const float* T_row, const float* delta_row, const int32_t* cidx_row, const void* color_out,
                      const void* logits, const void* feat, const void* sigma_out, const float* g_image, const float* g_depth,
                      const float* g_sem, const float* g_feat, int32_t N, int32_t S1, int32_t S2, int32_t C, int32_t Cpad,
                      int32_t D, float bg, float density_scale, float* d_h0, void* d_color_out, void* d_logits, void* d_feat,
                      int32_t mask_feat /* 1: d_feat rows are zeroed where feat <= 0 (feat = a post-ReLU activation) */,
                      const float* dots_row /* optional [rows]: <logits_s, g_sem> + <f_s, g_feat> from aln_sem_heads_bwd(dots_row), in place of
                                               the logits / feat rows (NULL then, like d_logits / d_feat) */,
                      int32_t* found_inf, void* stream);

/* ---- loss: autolabel/trainer.py:72-92 (rgb MSE + depth L1 over depth > 0.01 + feature L1 + CE over labelled rays), ONE launch.
 * Writes per-ray output gradients (times *loss_scale), terms[0..5) = {rgb, depth, feature, semantic, total} (summed in a fixed
 * order: bit-reproducible) and counts[0..2) = {rays with valid depth, labelled rays}.  `terms` must hold aln_loss_terms_floats()
 * floats (scratch behind the five results), `counts` 4 int32 of which counts[2] (an arrival ticket the kernel resets itself) must
 * be zero before the first launch. */
int32_t aln_loss_terms_floats(void);
int aln_loss_fwd_bwd(const float* image, const float* depth, const float* semantic, const float* features, const float* gt_rgb,
                     const float* gt_depth, const int32_t* gt_sem, const float* gt_feat, int32_t N, int32_t C, int32_t D,
                     int32_t Cf, float w_rgb, float w_depth, float w_sem, float w_feat, const float* loss_scale, int32_t* counts,
                     float* g_image, float* g_depth, float* g_sem, float* g_feat, float* terms, void* stream);

/* ---- optimizer: torch.optim.Adam of scripts/train.py:50-63 + GradScaler step/update of autolabel/trainer.py:45-48.
 * ONE launch.  state_i = 16 int32 {applied steps, growth tracker, found_inf, (caller's), step count per block [4..4+n_blocks),
 * ..., [15] = arrival ticket (zero before the first launch; the kernel resets it)}, state_f = {loss scale, learning rate (> 0:
 * overrides `lr`)}, consts (optional) >= 4 + 2*n_blocks floats: the step constants, for inspection.  Skips the update (and backs the scale off) when found_inf is set; refreshes the
 * fp16 table shadow; zeroes the gradients.  Parameter blocks follow torch's per-tensor semantics: block_kind 1 (semantic_out)
 * is skipped when the batch has no labelled ray, kind 2 (semantic_features) when additionally there is no feature loss
 * (their gradient is None in the reference, so torch.optim.Adam leaves them and their step counters untouched). */
int aln_adam_step(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid, int64_t n_total,
                  int32_t* state_i, float* state_f, float* consts, float lr, float beta1, float beta2, float eps, float wd_net,
                  float growth, float backoff, int32_t growth_interval, int32_t n_blocks, const int64_t* block_end /*host*/,
                  const int32_t* block_kind /*host*/, int32_t feature_loss,
                  int32_t skip_grid /* 1: block 0 (the table) was updated by aln_encode_bwd_binned(adam); only its step counter advances */,
                  const int32_t* counts /*device, optional*/,
                  uint32_t* step_dev /*optional: += 1 once per call (the step counter of aln_raygen_train & co. under graph replay)*/, void* stream);
/* The same step with the TABLE's gradient taken from the fp16 payload of the data-parallel exchange instead of grads[0, n_grid)
 * (grid_wire_f16[i] = averaged gradient of flat element i, as aln_encode_bwd_binned_wire and the SUM all-reduce leave it; the caller
 * has watched it for non-finite halves with aln_grad_unpack_f16(grad = NULL)): bit for bit the step aln_grad_unpack_f16 + aln_adam_step
 * take, without the fp32 copy of the table gradient.  The MLP blocks' gradients are read from grads and cleared as usual.  Replicated
 * optimizer only; the table is parameter block 0, n_grid a multiple of 4 (ABI 7). */
int aln_adam_step_wire(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid, int64_t n_total,
                       int32_t* state_i, float* state_f, float* consts, float lr, float beta1, float beta2, float eps, float wd_net,
                       float growth, float backoff, int32_t growth_interval, int32_t n_blocks, const int64_t* block_end /*host*/,
                       const int32_t* block_kind /*host*/, int32_t feature_loss, const void* grid_wire_f16,
                       const int32_t* counts /*device, optional*/, uint32_t* step_dev /*optional*/, void* stream);
/* The same step for a rank that owns only the slices [range_lo[k], range_hi[k]) of the table (n_ranges <= 8, ascending, disjoint,
 * multiples of 4; the table must be parameter block 0 and n_grid a multiple of 4) plus the whole MLP block: the sharded optimizer
 * of the data-parallel engine (no counterpart in the reference, which has no multi-GPU path; the arithmetic per parameter is
 * aln_adam_step's).  m, v are COMPACT -- the owned slices back to back, then the n_total - n_grid MLP moments.  Gradients,
 * master parameters and the fp16 table outside the owned slices are not touched (the gradient is cleared by
 * aln_grad_pack_f16_clear on its way to the reduce-scatter; the table comes back by all-gather). */
int aln_adam_step_ranges(float* params, float* grads, float* m, float* v, void* table_f16, int64_t n_grid, int64_t n_total,
                         int32_t* state_i, float* state_f, float* consts, float lr, float beta1, float beta2, float eps, float wd_net,
                         float growth, float backoff, int32_t growth_interval, int32_t n_blocks, const int64_t* block_end /*host*/,
                         const int32_t* block_kind /*host*/, int32_t feature_loss, int32_t n_ranges, const int64_t* range_lo /*host*/,
                         const int64_t* range_hi /*host*/, const int32_t* counts /*device, optional*/, uint32_t* step_dev, void* stream);
int aln_cast_f16(const float* src, void* dst, int64_t n, void* stream);
int aln_cast_f32(const void* src_f16, float* dst, int64_t n, void* stream);
/* fp16 wire format of the data-parallel exchange of the hash-grid gradient block (the reference has no multi-GPU path; the
 * gradient it averages is what scripts/train.py:50-63's optimizer consumes): out = fp16(grad * mul), and back grad = fp32(in)
 * with *found_inf raised on a non-finite element (grad == NULL: watch only -- aln_adam_step_wire reads the halves itself).
 * Pointers 16-byte aligned. */
int aln_grad_pack_f16(const float* grad, int64_t n, float mul, void* out_f16, void* stream);
int aln_grad_unpack_f16(const void* in_f16, int64_t n, float* grad, int32_t* found_inf, void* stream);
/* staging for a reduce-scatter: as aln_grad_pack_f16, plus out[n, n_pad) = 0 and grad[0, n) cleared behind the read */
int aln_grad_pack_f16_clear(float* grad, int64_t n, int64_t n_pad, float mul, void* out_f16, void* stream);

/* ---- feature-map file: autolabel/dataset.py:438-441 reads features.hdf through h5py, whose LZF filter (id 32000,
 * scripts/compute_feature_maps.py:85) wraps liblzf.  HOST pointers; returns the number of bytes produced or -1. */
int64_t aln_lzf_decompress(const void* src /*host*/, size_t n_in, void* dst /*host*/, size_t n_out);

#ifdef __cplusplus
}
#endif
#endif
