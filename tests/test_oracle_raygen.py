"""Pin oracle/raygen_oracle.py against vectors made by the reference's own code
(tests/golden/make_fixtures.py -> autolabel/dataset.py, model_utils.py)."""
import json
import os
import random

import numpy as np

from oracle import raygen_oracle as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_f1_compute_direction_bit_exact(golden_dir):
    f = _load(golden_dir, 'raygen_f1.npz')
    for name in ['toy', 'r32', 'replica', 'halfres']:
        w, h, fx, fy, cx, cy = f[f'{name}_intr']
        d, n = R.compute_direction(f[f'{name}_R'], f[f'{name}_idx'], int(w), fx, fy, cx, cy, False)
        assert np.array_equal(n, f[f'{name}_norm']), name
        # rotation: reference goes through BLAS sgemv (may fuse); allow 1 ulp
        assert np.max(np.abs(d - f[f'{name}_dirs'])) <= 1.2e-7, name


def _replay(golden_dir, tag):
    f = _load(golden_dir, f'raygen_f2_{tag}.npz')
    sampler = R.IndexSampler()
    sampler.update(f['semantics'])
    np.random.seed(0)
    random.seed(0)
    w, h, fx, fy, cx, cy = f['intr']
    feats = f['features'] if 'features' in f.files else None
    fhw = tuple(f['feat_shape'][:2]) if feats is not None else None
    out = R.next_train(f['images'], f['depths'], f['semantics'], f['poses'][:, :3, 3],
                       np.ascontiguousarray(f['poses'][:, :3, :3]), f['pixel_indices'], sampler, int(w),
                       (fx, fy, cx, cy), 8192, features=feats, feat_hw=fhw, h=int(h))
    return f, out


def test_f2_next_train_plain(golden_dir):
    f, out = _replay(golden_dir, 'plain')
    for k in ['pixels', 'depth', 'semantic', 'rays_o', 'direction_norms']:
        assert np.array_equal(out[k], f[f'batch_{k}']), k
    assert np.max(np.abs(out['rays_d'] - f['batch_rays_d'])) <= 1.2e-7
    assert out['semantic'].dtype == f['batch_semantic'].dtype
    assert (out['semantic'] == -1).all()


def test_f2_next_train_labelled_with_features(golden_dir):
    f, out = _replay(golden_dir, 'labelled')
    for k in ['pixels', 'depth', 'semantic', 'rays_o', 'direction_norms', 'features']:
        assert np.array_equal(out[k], f[f'batch_{k}']), k
    assert np.max(np.abs(out['rays_d'] - f['batch_rays_d'])) <= 1.2e-7
    assert (out['semantic'] >= 0).any()  # class-weighted branch exercised


def test_f3_get_test(golden_dir):
    f2 = _load(golden_dir, 'raygen_f2_plain.npz')
    f = _load(golden_dir, 'raygen_f3.npz')
    w, h, fx, fy, cx, cy = f2['intr']
    pose = f['pose']
    t = R.get_test(f2['images'][0], f2['depths'][0], f2['semantics'][0], pose[:3, 3],
                   np.ascontiguousarray(pose[:3, :3]), int(w), int(h), (fx, fy, cx, cy))
    assert t['direction_norms'].shape == (int(w) * int(h), 1)  # the [H*W,1] quirk
    for k in ['pixels', 'rays_o', 'depth', 'semantic', 'direction_norms']:
        assert np.array_equal(np.asarray(t[k]), f[k]), k
    assert np.max(np.abs(t['rays_d'] - f['rays_d'])) <= 1.2e-7
    assert t['depth'].dtype == f['depth'].dtype


def test_f4_index_sampler_known_answers(golden_dir):
    """The reference's own unit-test expectations (test/test_sampling.py:8-56)."""
    f4 = json.load(open(os.path.join(golden_dir, 'raygen_f4.json')))
    sem = np.zeros((2, 10), int)
    s = R.IndexSampler()
    s.update(sem)
    assert s.has_semantics == f4['empty_has_semantics'] and len(s.classes) == f4['empty_n_classes']
    sem[0, 5], sem[0, 0], sem[1, 5] = 1, 2, 3
    s.update(sem)
    assert [int(c) for c in s.classes] == f4['classes']
    for c, d in f4['index'].items():
        for i, a in d.items():
            assert list(s.index[int(c)][int(i)]) == a
        assert np.allclose(s.image_weights[int(c)], f4['image_weights'][c])
    assert s.sample_class() in [1, 2, 3]
    img, idx = s.sample(1, 1)
    assert img == 0 and idx[0] == 5
    img, idx = s.sample(2, 1)
    assert img == 0 and idx[0] == 0
    img, idx = s.sample(3, 5)
    assert img == 1 and len(idx) == 5 and (idx == 5).all()
    sem5 = np.zeros((5, 10), int)
    sem5[0, 5], sem5[2, 0], sem5[4, 5] = 1, 2, 3
    s5 = R.IndexSampler()
    s5.update(sem5)
    assert s5.semantic_indices() == f4['semantic_indices'] == [0, 2, 4]


def test_f5_convert_pose(golden_dir):
    f = _load(golden_dir, 'raygen_f5.npz')
    for T, want in zip(f['T_CW'], f['T_out']):
        got = R.convert_pose(T)
        assert np.array_equal(got, want.astype(np.float32))
        # net effect = row permutation (y,z,x) of T_WC (evaluation.py:453-457)
        T_WC = np.linalg.inv(T)
        assert np.allclose(got[:3, :3], T_WC[[1, 2, 0], :3], atol=1e-6)
        assert np.allclose(got[:3, 3], T_WC[[1, 2, 0], 3], atol=1e-6)
