"""Flags, model construction and checkpoint / params.pkl IO (interface of autolabel/model_utils.py)."""
import argparse
import glob
import os
import pickle

import torch

from .models import ALNetwork


def load_checkpoint(model, checkpoint_dir, device='cuda:0', reference=False):
    """Prefer '*best.pth', else the last checkpoint in sorted order; weights live under ['model'].
    reference=True: the checkpoint was written by the reference (tcnn modules) -- see import_reference_state_dict."""
    paths = sorted(glob.glob(f'{checkpoint_dir}/*.pth'))
    best = [p for p in paths if 'best.pth' in p]
    path = best[0] if best else paths[-1]
    sd = torch.load(path, map_location=device, weights_only=False)['model']
    if reference:
        import_reference_state_dict(model, sd)
    else:
        model.load_state_dict(sd)
    return model


# buffers the fork's NeRFRenderer registers that have no counterpart here (or only with cuda_ray)
_REFERENCE_ONLY_KEYS = ('density_grid', 'density_bitfield', 'step_counter', 'mean_density', 'iter_density', 'mean_count', 'local_step')


def import_reference_state_dict(model, sd):
    """Load `checkpoints/*.pth['model']` of a scene trained by the REFERENCE (autolabel/model_utils.py:9-18) -- SURVEY 8f N2.

    The reference's modules are tinycudann objects whose `params` are flat tensors; the layout they use is the one this
    build's master buffer was designed after, so the import is a checked reshape:
      * `encoder.grid_encoding.params`: per level `min(ceil8(res^3), 2^19)` entries x 2 features, levels concatenated;
      * `<net>.params` (FullyFusedMLP / CutlassMLP): the weight matrices in layer order, each [out, in] row-major, the first
        layer's `in` and the last layer's `out` padded to multiples of 16 (ones-padding on the input, zero rows on the output);
      * `encoder.encoder.params`, `encoder_dir.params` (Frequency, SphericalHarmonics): empty.
    Tensors may be fp16 (tcnn keeps an fp16 copy next to the fp32 master) and 1-D or 2-D.  Keys that only exist in the fork's
    renderer (density grid of the cuda_ray path, counters) are ignored; a size mismatch raises with both element counts.
    The grid position is switched to the fused multiply-add tcnn evaluates (`ALNetwork.set_tcnn_fma`).
    Parity unpinned: no tcnn checkpoint exists offline to pin the packing against (Oracle)."""
    own = model.state_dict()
    take, ignored = {}, []
    for k, v in sd.items():
        if k not in own:
            if k.split('.')[-1] in _REFERENCE_ONLY_KEYS or k in ('aabb_train', 'aabb_infer'):
                ignored.append(k)
                continue
            raise KeyError(f'reference checkpoint has {k!r}, which this model ({model.encoding}) has no slot for')
        t = v.detach()
        if k.endswith('.params'):
            t = t.float().reshape(-1)
            if t.numel() != own[k].numel():
                raise ValueError(f'{k}: checkpoint holds {t.numel()} parameters, this model expects {own[k].numel()} '
                                 '(tcnn packing: [out, in] matrices, in / out padded to 16; grid: min(ceil8(res^3), 2^19) x 2 per level)')
        take[k] = t.to(own[k].dtype).reshape(own[k].shape)
    missing = [k for k, v in own.items() if k.endswith('.params') and v.numel() and k not in take]
    if missing:
        raise KeyError(f'reference checkpoint lacks {missing}')
    model.load_state_dict(take, strict=False)
    model.set_tcnn_fma(True)
    return ignored


def model_flag_parser():
    parser = argparse.ArgumentParser()
    add = parser.add_argument
    add('--lr', type=float, default=5e-3)
    add('--geometric-features', '-g', type=int, default=15)
    add('--encoding', default='hg+freq', choices=['freq', 'hg', 'hg+freq'], type=str, help='Network positional encoding to use.')
    add('--features', type=str, default=None, choices=[None, 'fcn50', 'dino', 'lseg'], help='Use semantic feature supervision.')
    add('--rgb-weight', default=1.0, type=float)
    add('--semantic-weight', default=1.0, type=float)
    add('--feature-weight', default=0.5, type=float)
    add('--depth-weight', default=0.1, type=float)
    add('--feature-dim', default=64, type=int)
    return parser


def model_hash(flags):
    feats = flags.features if flags.features is not None else 'plain'
    return (f'g{flags.geometric_features}_{flags.encoding}_{feats}_rgb{flags.rgb_weight}_d{flags.depth_weight}'
            f'_s{flags.semantic_weight}_f{flags.feature_weight}')


def model_dir(scene_path, flags):
    if flags.workspace is None:
        return os.path.join(scene_path, 'nerf', model_hash(flags))
    scene_name = os.path.basename(os.path.normpath(flags.scene))
    return os.path.join(flags.workspace, scene_name, model_hash(flags))


def create_model(min_bounds, max_bounds, n_classes, flags, cuda_ray=False, **renderer_kwargs):
    """autolabel/model_utils.py:61-74 (which hard-codes cuda_ray=False).  cuda_ray=True switches the renderer to occupancy-grid
    marching (csrc/march.hip); renderer_kwargs: march_samples, max_steps, grid_size, density_thresh."""
    extents = max_bounds - min_bounds
    bound = (extents - (min_bounds + max_bounds) * 0.5).max()
    return ALNetwork(num_layers=2, num_layers_color=2, hidden_dim_color=128, hidden_dim=128,
                     geo_feat_dim=flags.geometric_features, encoding=flags.encoding, bound=float(bound),
                     hidden_dim_semantic=flags.feature_dim, cuda_ray=bool(cuda_ray), density_scale=1, semantic_classes=n_classes,
                     **renderer_kwargs)


def read_params(workspace):
    with open(os.path.join(workspace, 'params.pkl'), 'rb') as f:
        return pickle.load(f)


def write_params(workspace, flags):
    os.makedirs(workspace, exist_ok=True)
    with open(os.path.join(workspace, 'params.pkl'), 'wb') as f:
        pickle.dump(flags, f)
