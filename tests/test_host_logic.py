"""CPU tests of the host-side mirror of the reference interface and of the C-ABI library surface."""
import ctypes
import json
import os
import random
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from autolabel_amd import hip
    L = hip.lib()
    assert L.aln_abi_version() == hip.ABI_VERSION == 9
    header = open(os.path.join(ROOT, 'include', 'autolabel_hip.h')).read()
    declared = set(re.findall(r'\b(aln_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations found'
    for sym in declared | set(hip.declared_symbols()):
        assert hasattr(L, sym), f'{sym} declared but not exported'
    assert set(hip.declared_symbols()) - {'aln_last_error'} <= declared, 'binding uses symbols missing from include/autolabel_hip.h'


def test_the_drivers_build_check_runs_clean():
    """__graft_entry__.build() is what the driver calls on CPU every round: it must pass at the tree's own ABI version (it pinned a stale
    number for half of round 6 while the library, the bindings and the header had moved on)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    entry = importlib.import_module('__graft_entry__')
    entry.build()


def test_grid_desc_host_vs_c_and_error_reporting():
    from autolabel_amd import hip
    g = hip.make_grid_desc()
    g2 = hip.AlnGridDesc(16, 2, 19, 16, 2.0)
    assert hip.lib().aln_grid_desc_init(ctypes.byref(g2)) == 0
    for i in range(16):
        assert (g.scale[i], g.res[i], g.size[i], g.offset[i], g.dense[i]) == (g2.scale[i], g2.res[i], g2.size[i], g2.offset[i], g2.dense[i])
    assert g.n_entries * 2 == 14229504
    bad = hip.AlnGridDesc(99, 2, 19, 16, 2.0)
    assert hip.lib().aln_grid_desc_init(ctypes.byref(bad)) != 0 and b'n_levels' in hip.lib().aln_last_error()
    with pytest.raises(NotImplementedError):
        hip.make_enc_desc('bogus', 1.0)


def test_product_dataset_replays_reference_batches(golden_dir):
    """autolabel_amd.dataset.BaseDataset._next_train/_get_test vs the reference's own outputs (F2/F3)."""
    from autolabel_amd import dataset as D
    from autolabel_amd.utils import Camera
    for tag in ['plain', 'labelled']:
        f = np.load(os.path.join(golden_dir, f'raygen_f2_{tag}.npz'))
        w, h, fx, fy, cx, cy = f['intr']
        ds = D.BaseDataset(8192, Camera(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]]), (int(w), int(h))))
        ds.n_examples, ds.images, ds.depths, ds.semantics = 6, f['images'], f['depths'], f['semantics']
        ds.index_sampler.update(f['semantics'])
        ds.pixel_indices, ds.poses = f['pixel_indices'], f['poses']
        ds.rotations, ds.origins = np.ascontiguousarray(f['poses'][:, :3, :3]), f['poses'][:, :3, 3]
        if 'features' in f.files:
            Hf, Wf, Cf = f['feat_shape']
            ds._set_feature_maps(f['features'].reshape(6, Hf, Wf, Cf))
        np.random.seed(0); random.seed(0)
        b = ds._next_train()
        for k in [k[6:] for k in f.files if k.startswith('batch_')]:
            if k == 'rays_d':
                assert np.max(np.abs(b[k] - f['batch_rays_d'])) <= 1.2e-7
            else:
                assert np.array_equal(b[k], f[f'batch_{k}']), (tag, k)
                assert b[k].dtype == f[f'batch_{k}'].dtype, (tag, k)
        for T, P in zip(f['T_CW'], f['poses']):
            assert np.array_equal(D.convert_pose(T).astype(np.float32), P)
        if tag == 'plain':
            f3 = np.load(os.path.join(golden_dir, 'raygen_f3.npz'))
            t = ds._get_test(0)
            for k in ['pixels', 'rays_o', 'depth', 'semantic', 'direction_norms']:
                assert np.array_equal(np.asarray(t[k]), f3[k]), k
            assert t['H'] == 32 and t['W'] == 32


def test_index_sampler_reference_unit_tests(golden_dir):
    """The three cases of the reference's test/test_sampling.py against the product class."""
    from autolabel_amd.dataset import IndexSampler
    f4 = json.load(open(os.path.join(golden_dir, 'raygen_f4.json')))
    sem = np.zeros((2, 10), int)
    s = IndexSampler(); s.update(sem)
    assert not s.has_semantics and len(s.classes) == 0
    sem[0, 5], sem[0, 6] = 1, 2
    s.update(sem)
    assert list(s.classes) == [1, 2]
    sem = np.zeros((2, 10), int); sem[0, 5], sem[0, 0], sem[1, 5] = 1, 2, 3
    s = IndexSampler(); s.update(sem)
    assert [int(c) for c in s.classes] == f4['classes']
    assert s.sample_class() in [1, 2, 3]
    for cls, img, px in [(1, 0, 5), (2, 0, 0)]:
        i, idx = s.sample(cls, 1)
        assert i == img and idx[0] == px
    i, idx = s.sample(3, 5)
    assert i == 1 and len(idx) == 5 and np.random.choice(idx) == 5 and s.has_semantics
    for c, w in f4['image_weights'].items():
        assert np.allclose(s.image_weights[int(c)], w)
    sem = np.zeros((5, 10), int); sem[0, 5], sem[2, 0], sem[4, 5] = 1, 2, 3
    s = IndexSampler(); s.update(sem)
    assert s.semantic_indices() == f4['semantic_indices']


def test_model_utils_fixtures(golden_dir, tmp_path):
    from autolabel_amd import model_utils as mu
    p = mu.model_flag_parser(); p.add_argument('scene'); p.add_argument('--workspace', default=None)
    f6 = json.load(open(os.path.join(golden_dir, 'model_utils_f6.json')))
    for tag, v in f6.items():
        fl = p.parse_args(v['argv'])
        assert mu.model_hash(fl) == v['hash'] and mu.model_dir(fl.scene, fl) == v['dir']
        assert {k: getattr(fl, k) for k in v['flags']} == v['flags']
    f7 = json.load(open(os.path.join(golden_dir, 'model_utils_f7.json')))
    fl = p.parse_args(['/data/scene1', '--feature-dim', '64'])
    for case in f7:
        m = mu.create_model(np.array(case['min']), np.array(case['max']), 7, fl)
        kw = case['kwargs']
        assert m.bound == kw['bound'] and m.hidden_dim == kw['hidden_dim'] and m.hidden_dim_color == kw['hidden_dim_color']
        assert m.num_layers == kw['num_layers'] and m.num_layers_color == kw['num_layers_color'] and m.geo_feat_dim == kw['geo_feat_dim']
        assert m.semantic_classes == kw['semantic_classes'] and m.cuda_ray is False and m.density_scale == kw['density_scale']
    mu.write_params(str(tmp_path), fl)
    assert vars(mu.read_params(str(tmp_path))) == vars(fl)


def test_model_surface_and_parameter_groups():
    from autolabel_amd.models import ALNetwork
    m = ALNetwork(encoding='hg+freq', num_layers=2, hidden_dim=128, geo_feat_dim=15, num_layers_color=2, hidden_dim_color=128,
                  hidden_dim_semantic=64, semantic_classes=2, bound=2.0, cuda_ray=False, density_scale=1)
    for attr in ['encoder', 'sigma_net', 'color_net', 'semantic_features', 'semantic_out', 'encoder_dir', 'bound', 'cuda_ray', 'bg_radius']:
        assert hasattr(m, attr)
    assert m.in_dim == 44 and sum(p.numel() for p in m.encoder.parameters()) == 14229504
    assert sum(p.numel() for p in m.network_parameters()) == 24576 + 22528 + 9216 + 6144
    assert len(m.get_params(1e-3)) == 6 and m.mark_untrained_grid(None, None) is None
    g = m.encoder.grid_encoding.params
    assert g.abs().max() <= 1e-4 and g.std() > 1e-5
    opt = torch.optim.Adam([{'params': list(m.encoder.parameters())}, {'params': m.network_parameters(), 'weight_decay': 1e-6}], lr=5e-3)
    assert len(opt.param_groups) == 2


def test_ema_and_scheduler_cadence_follow_reference():
    """ema.update() and scheduler.step() run once per train_iterations call (autolabel/trainer.py:50-52)."""
    from autolabel_amd.trainer import ExponentialMovingAverage
    p = torch.nn.Parameter(torch.ones(3))
    ema = ExponentialMovingAverage([p], 0.95)
    with torch.no_grad():
        p.add_(1.0)
    ema.update()
    d = min(0.95, 2 / 11)
    assert torch.allclose(ema.shadow[0], torch.ones(3) + (1 - d) * 1.0)


def test_synthetic_scene_round_trips_through_scene_directory(tmp_path):
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import ArrayDataset, SceneDataset
    from autolabel_amd.utils import write_scene
    scene = synthetic.make_cube_scene(n_frames=4)
    write_scene(scene, str(tmp_path / 's'))
    ds = SceneDataset('train', str(tmp_path / 's'), factor=1.0, batch_size=1024)
    ref = ArrayDataset(scene, batch_size=1024)
    assert ds.n_examples == 4 and ds.n_classes == 3 and ds.images.shape == ref.images.shape
    assert np.array_equal(ds.depths, ref.depths) and np.array_equal(ds.semantics, ref.semantics)
    assert np.abs(ds.images - ref.images).max() <= 0.5 / 255 + 1e-6 and np.allclose(ds.poses, ref.poses, atol=1e-6)
    np.random.seed(0); random.seed(0)
    b = ds._next_train()
    assert b['rays_o'].shape == (1024, 3) and b['semantic'].dtype == np.int64 and (b['semantic'] >= -1).all()


def test_depth_frames_are_resized_like_the_references_cv2_call(tmp_path):
    """The reference's eager loader calls cv2.resize(depth, size, cv2.INTER_NEAREST) -- the flag lands in the `dst` slot, so depth is
    resized with cv2's DEFAULT, INTER_LINEAR (autolabel/dataset.py:391-393).  Known answers of OpenCV's published algorithm (cv2 is not
    in this image: restated, parity unpinned): exact factor 2 = the fast area path (a + b + c + d + 2) >> 2, round half UP; any other
    scale = pixel-centre bilinear in fp32 with clamped borders, saturate_cast = round half to EVEN."""
    from autolabel_amd.dataset import SceneDataset, _resize_linear
    from autolabel_amd import synthetic
    from autolabel_amd.utils import write_scene
    a = np.array([[0, 1, 10, 20], [1, 1, 30, 40], [65535, 65535, 2, 3], [65535, 65535, 4, 5]], dtype=np.uint16)
    # sums 3, 100, 262140, 14 -> 0.75 -> 1, 25, 65535, 3.5 -> 4 (half up; half-to-even would give 4 as well, 2.5 would not:)
    assert _resize_linear(a, (2, 2)).tolist() == [[1, 25], [65535, 4]]
    assert _resize_linear(np.array([[1, 2], [3, 4]], dtype=np.uint16), (1, 1)).tolist() == [[3]]          # 2.5 -> 3 (half up)
    # 5 -> 3 per axis: taps (0, 1/3), (2, 0), (3, 2/3); values 100 * (5 y + x)
    b = (np.arange(25, dtype=np.uint16).reshape(5, 5) * 100)
    assert _resize_linear(b, (3, 3)).tolist() == [[200, 367, 533], [1033, 1200, 1367], [1867, 2033, 2200]]
    # upsampling clamps at both borders (weight 0 beyond the last source pixel), half to even in the interior: 2 -> 4 of [0, 5]:
    # fx = -0.25 -> (0, 0) ; 0.25 -> 1.25 -> 1 ; 0.75 -> 3.75 -> 4 ; 1.25 -> clamped (1, 0) -> 5
    assert _resize_linear(np.array([[0, 5]], dtype=np.uint16), (4, 1)).tolist() == [[0, 1, 4, 5]]
    assert _resize_linear(a, (4, 4)).tolist() == a.tolist()
    # through the loader: factor 2 halves a 32 x 32 scene; every supervised depth is the rounded mean of its 2 x 2 block
    scene = synthetic.make_cube_scene(n_frames=2)
    write_scene(scene, str(tmp_path / 's'))
    full = SceneDataset('train', str(tmp_path / 's'), factor=1.0, batch_size=512)
    half = SceneDataset('train', str(tmp_path / 's'), factor=2.0, batch_size=512)
    d = full.depths.reshape(2, 32, 32).astype(np.uint32)
    want = (d[:, 0::2, 0::2] + d[:, 0::2, 1::2] + d[:, 1::2, 0::2] + d[:, 1::2, 1::2] + 2) >> 2
    assert half.depths.shape[1:] == (16, 16) or half.depths.reshape(2, -1).shape[1] == 256
    assert np.array_equal(half.depths.reshape(2, 16, 16), want)
    # rgb stays nearest (the reference passes interpolation= by keyword there: autolabel/dataset.py:369-371)
    assert np.array_equal(half.images.reshape(2, 16, 16, 3), full.images.reshape(2, 32, 32, 3)[:, 0::2, 0::2])


def test_lazy_loader_divides_every_frame_by_255_like_the_reference(tmp_path):
    """autolabel/dataset.py:63-70: LazyImageLoader converts whatever it opens to float32 / 255 -- depth frames too -- and resizes with
    the interpolation it was given (nearest, by keyword: dataset.py:395-400)."""
    from PIL import Image
    from autolabel_amd.dataset import LazyImageLoader
    d = (np.arange(16, dtype=np.uint16).reshape(4, 4) * 1000)
    Image.fromarray(d).save(str(tmp_path / 'd.png'))
    got = LazyImageLoader([str(tmp_path / 'd.png')], (2, 2))[0]
    assert got.dtype == np.float32 and np.array_equal(got, (d.astype(np.float32) / 255.)[0::2, 0::2])


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    """`--gpus N` must mean N ranks.  Under a launcher (RANK set) a mismatch exits non-zero before any GPU call;
    with no launcher bench.py starts the ranks itself (covered on the GPU box by the driver's --gpus runs)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr
