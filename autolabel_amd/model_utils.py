"""Flags, model construction and checkpoint / params.pkl IO (interface of autolabel/model_utils.py)."""
import argparse
import glob
import os
import pickle

import torch

from .models import ALNetwork


def load_checkpoint(model, checkpoint_dir, device='cuda:0'):
    """Prefer '*best.pth', else the last checkpoint in sorted order; weights live under ['model']."""
    paths = sorted(glob.glob(f'{checkpoint_dir}/*.pth'))
    best = [p for p in paths if 'best.pth' in p]
    path = best[0] if best else paths[-1]
    model.load_state_dict(torch.load(path, map_location=device)['model'])
    return model


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


def create_model(min_bounds, max_bounds, n_classes, flags):
    extents = max_bounds - min_bounds
    bound = (extents - (min_bounds + max_bounds) * 0.5).max()
    return ALNetwork(num_layers=2, num_layers_color=2, hidden_dim_color=128, hidden_dim=128,
                     geo_feat_dim=flags.geometric_features, encoding=flags.encoding, bound=float(bound),
                     hidden_dim_semantic=flags.feature_dim, cuda_ray=False, density_scale=1, semantic_classes=n_classes)


def read_params(workspace):
    with open(os.path.join(workspace, 'params.pkl'), 'rb') as f:
        return pickle.load(f)


def write_params(workspace, flags):
    os.makedirs(workspace, exist_ok=True)
    with open(os.path.join(workspace, 'params.pkl'), 'wb') as f:
        pickle.dump(flags, f)
