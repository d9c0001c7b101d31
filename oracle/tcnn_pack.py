"""Writer of a reference-format checkpoint from the ORACLE's own per-layer matrices.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **Parity unpinned**: tinycudann is absent from /root/reference, so this
restates its published parameter packing; it exists so that the checkpoint importer (autolabel_amd/model_utils.py:
import_reference_state_dict; loader call site autolabel/model_utils.py:9-18) is tested against a file that was
NOT produced by the model under test.

Packing restated (tiny-cuda-nn, ``Network`` / ``Encoding`` modules of the torch bindings, ``module.params`` = one flat tensor):

* ``FullyFusedMLP`` / ``CutlassMLP``: the weight matrices in layer order, each [out, in] row-major; the first layer's ``in`` and
  the last layer's ``out`` are padded to multiples of 16 (the padded input columns multiply the constant-one padding of the
  input, the padded output rows are dead).  No biases.
* ``GridEncoding``: levels concatenated; level l holds ``min(ceil8(res_l^3), 2^log2_hashmap_size)`` entries of
  ``n_features_per_level`` values, entry-major.
* ``Frequency`` / ``SphericalHarmonics``: no parameters (empty tensor).
* the bindings keep ``params`` in fp32 by default; half-precision copies (``.half()`` checkpoints) occur and tensors may be saved
  1-D (flat) or with their natural 2-D shape.
"""
import torch

from . import nerf_oracle as O

# state-dict prefixes of autolabel/models.py:84-136 for the oracle's head names
HEAD_KEYS = {'sigma': 'sigma_net', 'color': 'color_net', 'semf': 'semantic_features', 'semo': 'semantic_out'}


def pack_reference_state_dict(params, cfg, grid_dtype=torch.float16, mlp_dtype=torch.float32, two_d=('color',), bound=1.0):
    """params: the oracle's dict (``grid`` [n_entries, F]; ``<head>.<layer>`` [out_pad, in_pad]) -> ``checkpoint['model']``."""
    sd = {}
    if cfg.encoding != 'freq':
        g = params['grid'].detach().reshape(-1).to(grid_dtype)
        sd['encoder.grid_encoding.params'] = g
    if cfg.encoding != 'hg':
        sd['encoder.encoder.params'] = torch.zeros(0)         # tcnn Frequency: no parameters
    sd['encoder_dir.params'] = torch.zeros(0)                 # tcnn SphericalHarmonics: no parameters
    shapes = O.mlp_shapes(cfg)
    for head, key in HEAD_KEYS.items():
        mats = [params[f'{head}.{i}'].detach() for i in range(len(shapes[head]))]
        for m, (o, i) in zip(mats, shapes[head]):
            assert tuple(m.shape) == (o, i), (head, m.shape, (o, i))
        flat = torch.cat([m.reshape(-1) for m in mats]).to(mlp_dtype)
        sd[f'{key}.params'] = flat.reshape(1, -1) if head in two_d else flat     # 1-D and 2-D tensors both occur
    # buffers of the fork's NeRFRenderer that travel with every checkpoint
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    sd['aabb_train'], sd['aabb_infer'] = aabb, aabb.clone()
    sd['density_grid'] = torch.zeros(1, 128 ** 3)
    sd['density_bitfield'] = torch.zeros(128 ** 3 // 8, dtype=torch.uint8)
    sd['step_counter'] = torch.zeros(16, 2, dtype=torch.int32)
    sd['mean_density'] = torch.zeros(())
    return sd
