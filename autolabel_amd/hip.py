"""ctypes binding of libautolabel_hip.so (C ABI: include/autolabel_hip.h).

Fails loudly: there is no CPU fallback for the hot path.
"""
import ctypes as C
import os

import numpy as np
import torch

from .build import LIB

ALN_MAX_LEVELS = 16
ABI_VERSION = 9   # include/autolabel_hip.h: ALN_ABI_VERSION
vp, i32, u32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64, C.c_float, C.c_double


class AlnGridDesc(C.Structure):
    _fields_ = [('n_levels', i32), ('n_features', i32), ('log2_hashmap_size', i32), ('base_resolution', i32),
                ('per_level_scale', f32), ('scale', f32 * ALN_MAX_LEVELS), ('res', u32 * ALN_MAX_LEVELS),
                ('size', u32 * ALN_MAX_LEVELS), ('offset', u32 * ALN_MAX_LEVELS), ('dense', u32 * ALN_MAX_LEVELS),
                ('n_entries', u32), ('pos_fma', i32)]


class AlnEncDesc(C.Structure):
    _fields_ = [('n_freq', i32), ('freq_normalized', i32), ('use_grid', i32), ('enc_dim', i32), ('enc_pad', i32),
                ('bound', f32), ('grid', AlnGridDesc)]


class AlnMlpDesc(C.Structure):
    _fields_ = [('in_pad', i32), ('hidden', i32), ('out_pad', i32), ('n_hidden', i32), ('wf', vp), ('wb', vp), ('wr', vp), ('dw_ws', vp),
                ('dw_ws_bytes', i64), ('defer_dw_reduce', i32), ('x_tiled', i32), ('x_pitch', i64)]


class AlnAdamFuse(C.Structure):
    _fields_ = [('params', vp), ('m', vp), ('v', vp), ('table_f16', vp), ('state_i', vp), ('state_f', vp), ('lr', f32), ('beta1', f32),
                ('beta2', f32), ('eps', f32)]


class AlnFrames(C.Structure):
    _fields_ = [('images', vp), ('depths', vp), ('semantics', vp), ('features', vp), ('rotations', vp), ('origins', vp),
                ('pixel_indices', vp), ('n_frames', i32), ('w', i32), ('h', i32), ('n_pix', i32), ('feat_w', i32),
                ('feat_h', i32), ('feat_c', i32), ('fx', f64), ('fy', f64), ('cx', f64), ('cy', f64), ('cls_offsets', vp),
                ('cls_pixels', vp), ('n_classes', i32), ('sem_ratio', f32)]


class AlnBatch(C.Structure):
    _fields_ = [('rays_o', vp), ('rays_d', vp), ('norms', vp), ('pixels', vp), ('depth', vp), ('semantic', vp),
                ('features', vp)]


_SIGS = {
    'aln_abi_version': (i32, []),
    'aln_grid_desc_init': (i32, [vp]),
    'aln_compute_direction': (i32, [vp, vp, i32, i32, f64, f64, f64, f64, vp, vp, vp, vp]),
    'aln_raygen_train': (i32, [vp, vp, i32, i32, i32, i32, u32, u32, vp, vp, vp, vp, vp]),
    'aln_raygen_frame': (i32, [vp, vp, i32, vp]),
    'aln_ray_aabb': (i32, [vp, vp, i32, f32, f32, vp, vp, vp]),
    'aln_sh4': (i32, [vp, i32, i32, vp, vp]),
    'aln_sample_coarse': (i32, [vp, vp, i32, i32, f32, f32, i32, u32, u32, vp, vp, vp, vp, vp, vp]),
    'aln_sample_fine': (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, u32, u32, vp, vp, vp, vp]),
    'aln_encode_fwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'aln_encode_fwd_ws_bytes': (i64, [vp, i32]),
    'aln_encode_fwd_phased': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'aln_encode_fwd_planes': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, i64, vp]),
    'aln_encode_fwd_cells': (i32, [vp, vp, i32, u32, u32, vp, i32, i32, vp, vp, vp]),
    'aln_encode_bwd_binned_ws_bytes': (i64, [vp, i32]),
    'aln_encode_bwd_binned_tile_rows': (i32, []),
    'aln_encode_bwd_binned': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    'aln_encode_bwd_binned_wire': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, vp, f32, vp]),
    'aln_encode_bwd_binned_phase': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, f32, i32, vp]),
    'aln_wide_nt': (i32, [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp, i32, i32, vp, i32, vp, i32, vp, vp]),
    'aln_wide_tn_ws_bytes': (i64, [i32, i32, i32]),
    'aln_wide_tn': (i32, [vp, i32, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, vp, vp]),
    'aln_transpose_f16': (i32, [vp, i32, i32, vp, vp]),
    'aln_wide_nt_gen': (i32, [vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp, vp]),
    'aln_composite_out_featsums': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    'aln_wide_nt_maskgen': (i32, [vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp]),
    'aln_wide_tn_gen': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp]),
    'aln_wide_tn_din_ws_bytes': (i64, [i32, i32]),
    'aln_wide_tn_din': (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    'aln_mlp_repack': (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    'aln_mlp_rowmajor_halves': (i64, [i32, i32, i32, i32]),
    'aln_mlp_repack_all': (i32, [i32, vp, vp, vp]),
    'aln_mlp_has_recompute': (i32, [i32, i32, i32, i32]),
    'aln_mlp_supports_tiled': (i32, [i32, i32, i32, i32]),
    'aln_mlp_bwd_blocks': (i32, [vp, i32]),
    'aln_mlp_dw_reduce_all': (i32, [i32, vp, vp, vp, vp]),
    'aln_mlp_dw_reduce_slabs': (i32, [i32, vp, vp, vp, vp, vp]),
    'aln_sem_heads_bwd_slabs': (i32, [vp, vp, i32, i32, i32]),
    'aln_mlp_dw_ws_bytes': (i64, [i32, i32, i32, i32]),
    'aln_mlp_frag_halves': (i64, [i32, i32, i32, i32, i32]),
    'aln_mlp_fwd': (i32, [vp, vp, i32, vp, vp, vp, vp, vp]),
    'aln_density_fwd': (i32, [vp, vp, i32, vp, vp, vp, vp, vp]),
    'aln_mlp_bwd': (i32, [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
    'aln_mlp_bwd_dso': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    'aln_sem_heads_fwd': (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    'aln_sem_heads_fwd_sums': (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    'aln_sem_heads_bwd': (i32, [vp] * 7 + [i32] * 7 + [vp] * 4 + [i32] + [vp] * 3),
    'aln_sigma_act': (i32, [vp, i32, vp, vp]),
    'aln_compact_live_ws_ints': (i32, [i32]),
    'aln_compact_live': (i32, [vp, i32, f32, vp, vp, vp, vp, vp]),
    'aln_compact_live_color_in': (i32, [vp, i32, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, i32, vp, vp]),
    'aln_color_fwd': (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp]),
    'aln_build_color_in': (i32, [vp, vp, i32, vp, vp, i32, i32, i32, vp, i32, i32, vp, vp]),
    'aln_build_sem_in': (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    'aln_assemble_grads': (i32, [vp, vp, i32, vp, i32, i32, vp, i32, vp, i32, i32, vp, vp, vp]),
    'aln_assemble_dsemf_out': (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    'aln_composite_fwd': (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'aln_march_rays': (i32, [vp, vp, i32, i32, f32, f32, vp, i32, i32, i32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'aln_grid_points': (i32, [i32, f32, u32, u32, vp, vp, vp, vp]),
    'aln_grid_update': (i32, [vp, vp, i32, f32, f32, f32, vp, vp, vp, vp]),
    'aln_bitfield_count': (i32, [vp, i64, vp, vp]),
    'aln_mark_untrained_grid': (i32, [vp, i32, f32, vp, i32, f32, f32, f32, f32, f32, f32, f32, i32, vp]),
    'aln_composite_out': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    'aln_composite_bwd': (i32, [vp] * 16 + [i32] * 6 + [f32, f32] + [vp] * 4 + [i32] + [vp] * 3),
    'aln_loss_terms_floats': (i32, []),
    'aln_loss_fwd_bwd': (i32, [vp] * 8 + [i32] * 4 + [f32] * 4 + [vp] * 8),
    'aln_adam_step': (i32, [vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, i32, vp, vp, i32, i32, vp, vp, vp]),
    'aln_adam_step_wire': (i32, [vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, i32, vp, vp, i32, vp, vp, vp, vp]),
    'aln_similarity_argmax': (i32, [vp, i32, i32, vp, i32, vp, vp]),
    'aln_cast_f16': (i32, [vp, vp, i64, vp]),
    'aln_cast_f32': (i32, [vp, vp, i64, vp]),
    'aln_grad_pack_f16': (i32, [vp, i64, f32, vp, vp]),
    'aln_grad_pack_f16_clear': (i32, [vp, i64, i64, f32, vp, vp]),
    'aln_adam_step_ranges': (i32, [vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    'aln_grad_unpack_f16': (i32, [vp, i64, vp, vp, vp]),
    'aln_lzf_decompress': (i64, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
}

_lib = None


def lib():
    """Load the HIP library; raise (never fall back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f'{LIB} not found: build it with `python -m autolabel_amd.build` '
                               '(hipcc --offload-arch=gfx950). autolabel_amd has no CPU fallback.')
        L = C.CDLL(LIB)
        L.aln_abi_version.restype = C.c_int
        if L.aln_abi_version() != ABI_VERSION:   # a signature-only change would otherwise pass garbage arguments silently
            raise RuntimeError(f'{LIB} implements ABI {L.aln_abi_version()}, these bindings were written for ABI {ABI_VERSION}: '
                               'rebuild it with `python -m autolabel_amd.build`')
        from . import build as _build
        if LIB == _build.LIB and _build.needs_build():
            import warnings
            warnings.warn(f'{LIB} is older than its sources: rebuild it with `python -m autolabel_amd.build`')
        L.aln_last_error.restype = C.c_char_p
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def declared_symbols():
    return ['aln_last_error'] + list(_SIGS)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('autolabel_amd: no HIP device visible; the hot path has no CPU fallback')


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib().aln_last_error().decode()}')


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'need a contiguous device tensor'
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_grid_desc(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=2.0, pos_fma=False):
    """tcnn GridEncoding level table (autolabel/models.py:38-48), evaluated in fp32 on the host."""
    g = AlnGridDesc(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale)
    g.pos_fma = int(bool(pos_fma))
    offset = 0
    for l in range(n_levels):
        scale = np.float32(np.exp2(np.float32(l) * np.log2(np.float32(per_level_scale))) * np.float32(base_resolution)
                           - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        size = min((min(res ** 3, 0x7FFFFFFF) + 7) // 8 * 8, 1 << log2_hashmap_size)
        g.scale[l], g.res[l], g.size[l], g.offset[l], g.dense[l] = float(scale), res, size, offset, int(res ** 3 <= size)
        offset += size
    g.n_entries = offset
    return g


def make_enc_desc(encoding, bound, grid=None):
    """Encoder selection of ALNetwork._get_encoder (autolabel/models.py:138-148)."""
    if encoding not in ('freq', 'hg', 'hg+freq'):
        raise NotImplementedError(f'Unknown input encoding {encoding}')
    e = AlnEncDesc()
    e.n_freq = {'hg+freq': 2, 'freq': 10, 'hg': 0}[encoding]
    e.freq_normalized = int(encoding == 'freq')
    e.use_grid = int(encoding != 'freq')
    e.grid = grid if grid is not None else make_grid_desc()
    e.enc_dim = 6 * e.n_freq + (e.grid.n_levels * e.grid.n_features if e.use_grid else 0)
    e.enc_pad = (e.enc_dim + 15) // 16 * 16
    e.bound = bound
    return e
