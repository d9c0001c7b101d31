"""Generate golden vectors by IMPORTING the reference's Python (this container only).

    python tests/golden/make_fixtures.py        # writes tests/golden/*.npz / *.json

The reference (/root/reference) cannot travel to the GPU box, so its outputs on
seeded inputs are committed as data.  Modules the image lacks are stubbed:
``cv2``/``h5py`` (unused on the exercised path), ``numba`` (``njit`` = identity:
the decorated function is plain numpy) and ``torch_ngp.nerf.provider`` whose
``nerf_matrix_to_ngp`` is external to the reference tree -- restated below from
public upstream (ashawkey/torch-ngp nerf/provider.py); fixture F5 is therefore
pinned only up to that restatement (cross-checked against the permutation the
reference applies itself at autolabel/evaluation.py:453-457).

Fixtures:
  F1 _compute_direction   dataset.py:17-37      raygen_f1.npz
  F2 _next_train          dataset.py:182-242    raygen_f2.npz
  F3 _get_test            dataset.py:244-266    raygen_f3.npz
  F4 IndexSampler         test/test_sampling.py known answers -> raygen_f4.json
  F5 _convert_pose        dataset.py:268-274    raygen_f5.npz
  F6 model_hash/model_dir model_utils.py:43-58  model_utils_f6.json
  F7 create_model bound   model_utils.py:62-63  model_utils_f7.json
"""
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def _install_stubs():
    for name in ['cv2', 'h5py', 'tinycudann', 'tensorboardX']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['cv2'].INTER_CUBIC = 2
    sys.modules['cv2'].INTER_NEAREST = 0
    numba = types.ModuleType('numba')
    numba.njit = lambda f=None, **kw: f if f is not None else (lambda g: g)
    sys.modules['numba'] = numba

    def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0, 0, 0)):
        return np.array([
            [pose[1, 0], -pose[1, 1], -pose[1, 2], pose[1, 3] * scale + offset[0]],
            [pose[2, 0], -pose[2, 1], -pose[2, 2], pose[2, 3] * scale + offset[1]],
            [pose[0, 0], -pose[0, 1], -pose[0, 2], pose[0, 3] * scale + offset[2]],
            [0, 0, 0, 1],
        ], dtype=np.float32)

    names = ['torch_ngp', 'torch_ngp.nerf', 'torch_ngp.nerf.provider', 'torch_ngp.nerf.renderer',
             'torch_ngp.nerf.utils', 'torch_ngp.gridencoder', 'torch_ngp.encoding', 'torch_ngp.activation',
             'torch_ngp.ffmlp']
    for n in names:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules['torch_ngp.nerf.provider'].nerf_matrix_to_ngp = nerf_matrix_to_ngp
    import torch
    sys.modules['torch_ngp.nerf.renderer'].NeRFRenderer = type('NeRFRenderer', (torch.nn.Module,), {})
    sys.modules['torch_ngp.nerf.utils'].Trainer = type('Trainer', (), {})
    for n, attr in [('gridencoder', 'GridEncoder'), ('encoding', 'get_encoder'), ('activation', 'trunc_exp'),
                    ('ffmlp', 'FFMLP')]:
        setattr(sys.modules[f'torch_ngp.{n}'], attr, None)
    sys.path.insert(0, REF)


def random_rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    return q


def make_dataset(ds_mod, utils_mod, w, h, n_frames, fx, fy, cx, cy, rng, batch_size, labelled=(), features=None):
    """In-memory BaseDataset with seeded synthetic arrays (no files)."""
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    cam = utils_mod.Camera(K, (w, h))
    ds = ds_mod.BaseDataset(batch_size, cam)
    ds.n_examples = n_frames
    ds.images = rng.random((n_frames, h * w, 3)).astype(np.float32)
    ds.depths = rng.integers(0, 5000, size=(n_frames, h * w)).astype(np.uint16)
    ds.depths[:, ::7] = 0
    sem = np.zeros((n_frames, h * w), dtype=np.uint8)
    for f in labelled:
        sem[f, : (h * w) // 3] = 1
        sem[f, (h * w) // 3: (h * w) // 2] = 2
    ds.semantics = sem
    ds.index_sampler.update(sem)
    ds.pixel_indices = np.arange(h * w)[rng.random(h * w) > 0.1]
    poses = []
    T_CWs = []
    for _ in range(n_frames):
        T = np.eye(4)
        T[:3, :3] = random_rotation(rng)
        T[:3, 3] = rng.normal(size=3)
        T_CWs.append(T)
        poses.append(ds._convert_pose(T).astype(np.float32))
    ds.poses = np.stack(poses)
    ds.rotations = np.ascontiguousarray(ds.poses[:, :3, :3])
    ds.origins = ds.poses[:, :3, 3]
    if features is not None:
        Hf, Wf, C = features
        ds.features = rng.normal(size=(n_frames, Hf * Wf, C)).astype(np.float16)
        ds.feature_width, ds.feature_height, ds.feature_dim = Wf, Hf, C
        sf = np.array([Wf / w, Hf / h])
        ds._scale_to_feature_xy = lambda xy: (xy * sf).astype(int)
    return ds, np.stack(T_CWs)


def main():
    _install_stubs()
    from autolabel import dataset as ds_mod
    from autolabel import utils as utils_mod
    from autolabel import model_utils as mu

    # ---- F1
    f1 = {}
    rng = np.random.default_rng(1)
    cases = {
        'toy': (np.eye(3, dtype=np.float32), 4, 3, 2.0, 2.0, 1.5, 1.0),
        'r32': (random_rotation(rng).astype(np.float32), 32, 32, 16.0, 16.0, 15.5, 15.5),
        'replica': (random_rotation(rng).astype(np.float32), 640, 480, 320.0, 320.0, 319.5, 239.5),
        'halfres': (random_rotation(rng).astype(np.float32), 320, 240, 160.0, 160.0, 159.75, 119.75),
    }
    for name, (R, w, h, fx, fy, cx, cy) in cases.items():
        idx = np.arange(w * h)
        if name in ('replica', 'halfres'):
            idx = rng.integers(0, w * h, size=4096)
        d, n = ds_mod._compute_direction(R, idx, w, np.float64(fx), np.float64(fy), np.float64(cx),
                                         np.float64(cy), False)
        f1[f'{name}_R'] = R
        f1[f'{name}_idx'] = idx
        f1[f'{name}_intr'] = np.array([w, h, fx, fy, cx, cy], dtype=np.float64)
        f1[f'{name}_dirs'] = d
        f1[f'{name}_norm'] = n
    np.savez_compressed(os.path.join(HERE, 'raygen_f1.npz'), **f1)

    # ---- F2 / F3
    for tag, labelled, feats in [('plain', (), None), ('labelled', (1, 4), (8, 8, 5))]:
        rng = np.random.default_rng(2)
        ds, T_CWs = make_dataset(ds_mod, utils_mod, 32, 32, 6, 16.0, 16.0, 15.5, 15.5, rng, 8192, labelled, feats)
        np.random.seed(0)
        random.seed(0)
        # record the host RNG draws the reference makes so the restatement can replay them
        batch = ds._next_train()
        out = {f'batch_{k}': v for k, v in batch.items()}
        out.update(images=ds.images, depths=ds.depths, semantics=ds.semantics, pixel_indices=ds.pixel_indices,
                   poses=ds.poses, T_CW=T_CWs, intr=np.array([32, 32, 16.0, 16.0, 15.5, 15.5]))
        if feats is not None:
            out['features'] = ds.features
            out['feat_shape'] = np.array(feats)
        if tag == 'plain':
            t = ds._get_test(0)
            np.savez_compressed(os.path.join(HERE, 'raygen_f3.npz'),
                                **{k: np.asarray(v) for k, v in t.items()}, pose=ds.poses[0])
        np.savez_compressed(os.path.join(HERE, f'raygen_f2_{tag}.npz'), **out)

    # ---- F4: the reference's own unit-test expectations (test/test_sampling.py:8-56)
    f4 = {}
    s = ds_mod.IndexSampler()
    sem = np.zeros((2, 10), int)
    s.update(sem)
    f4['empty_has_semantics'] = bool(s.has_semantics)
    f4['empty_n_classes'] = int(len(s.classes))
    sem[0, 5] = 1
    sem[0, 0] = 2
    sem[1, 5] = 3
    s.update(sem)
    f4['classes'] = [int(c) for c in s.classes]
    f4['index'] = {str(c): {str(i): [int(x) for x in a] for i, a in d.items()} for c, d in s.index.items()}
    f4['image_weights'] = {str(c): [float(x) for x in w] for c, w in s.image_weights.items()}
    sem5 = np.zeros((5, 10), int)
    sem5[0, 5] = 1
    sem5[2, 0] = 2
    sem5[4, 5] = 3
    s5 = ds_mod.IndexSampler()
    s5.update(sem5)
    f4['semantic_indices'] = [int(i) for i in s5.semantic_indices()]
    json.dump(f4, open(os.path.join(HERE, 'raygen_f4.json'), 'w'), indent=1)

    # ---- F5
    rng = np.random.default_rng(5)
    cam = utils_mod.Camera(np.eye(3), (4, 4))
    ds = ds_mod.BaseDataset(512, cam)
    T_in, T_out = [], []
    for _ in range(3):
        T = np.eye(4)
        T[:3, :3] = random_rotation(rng)
        T[:3, 3] = rng.normal(size=3)
        T_in.append(T)
        T_out.append(ds._convert_pose(T))
    np.savez_compressed(os.path.join(HERE, 'raygen_f5.npz'), T_CW=np.stack(T_in), T_out=np.stack(T_out))

    # ---- F6 / F7
    parser = mu.model_flag_parser()
    parser.add_argument('scene')
    parser.add_argument('--workspace', default=None)
    f6 = {}
    for tag, argv in [('default', ['/data/scene1']), ('dino', ['/data/scene1/', '--features', 'dino']),
                      ('ws', ['/data/scene1/', '--workspace', '/tmp/ws', '--feature-dim', '512', '-g', '31'])]:
        fl = parser.parse_args(argv)
        f6[tag] = dict(argv=argv, hash=mu.model_hash(fl), dir=mu.model_dir(fl.scene, fl),
                       flags={k: v for k, v in vars(fl).items()})
    json.dump(f6, open(os.path.join(HERE, 'model_utils_f6.json'), 'w'), indent=1)

    f7 = []
    rng = np.random.default_rng(7)
    mu.ALNetwork = lambda **kw: kw  # capture the kwargs create_model passes (tcnn is absent)
    fl = parser.parse_args(['/data/scene1', '--feature-dim', '512'])
    for _ in range(3):
        lo = rng.normal(size=3) - 2
        hi = lo + rng.random(3) * 5 + 0.5
        kw = mu.create_model(lo, hi, 7, fl)
        f7.append(dict(min=lo.tolist(), max=hi.tolist(), kwargs=kw))
    json.dump(f7, open(os.path.join(HERE, 'model_utils_f7.json'), 'w'), indent=1)
    print('fixtures written to', HERE)


if __name__ == '__main__':
    main()
