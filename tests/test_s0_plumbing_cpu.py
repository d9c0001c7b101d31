"""BASELINE.json configs[0] -- the CPU-runnable plumbing case: 32x32 synthetic RGB-D cube, hash grid L = 4, host data path
(the product's mirror of `BaseDataset._next_train`, dataset.py:182-242) feeding the CPU oracle's render -> 4-term loss ->
Adam loop (trainer.py:54-94, scripts/train.py:50-63).  No GPU: this is the reference-side half of every parity test, run
end to end.  (Batch cut from 8192 to 1024 rays and 16+16 samples so the CPU suite stays within minutes.)"""
import random

import numpy as np
import torch

from oracle import nerf_oracle as O


def test_cube_scene_trains_on_the_cpu_oracle():
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import ArrayDataset
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    scene = synthetic.make_cube_scene(n_frames=12, size=32, seed=0)
    ds = ArrayDataset(scene, batch_size=1024)
    lo, hi = scene['min_bounds'], scene['max_bounds']
    bound = float(((hi - lo) - (lo + hi) * 0.5).max())           # autolabel/model_utils.py:62-63
    cfg = O.ModelConfig(encoding='hg+freq', feature_dim=64, n_classes=scene['n_classes'], bound=bound, grid=O.GridSpec(n_levels=4))
    model = O.OracleModel(cfg, seed=0)
    with torch.no_grad():
        model.params['grid'].mul_(1e3)
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in model.params.items()}
    g = torch.Generator().manual_seed(1)
    photometric, labelled = [], 0
    for it in range(16):
        b = ds._next_train()
        assert b['rays_o'].shape == (1024, 3) and b['rays_d'].dtype == np.float32 and b['semantic'].dtype == np.int64
        assert np.allclose(np.linalg.norm(b['rays_d'], axis=1), 1.0, atol=1e-5)
        labelled += int((b['semantic'] >= 0).sum())
        t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items() if isinstance(v, np.ndarray)}
        out = model.run(t['rays_o'], t['rays_d'], t['direction_norms'], 16, 16, perturb=True,
                        noise_coarse=torch.rand(1024, 16, generator=g), u_fine=torch.rand(1024, 16, generator=g))
        loss, terms = O.loss_fn(out, {'pixels': t['pixels'], 'depth': t['depth'], 'semantic': t['semantic']}, feature_loss=False)
        for p in model.params.values():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for k, p in model.params.items():
                if p.grad is not None:
                    O.adam_update(p, p.grad, state[k][0], state[k][1], it + 1, 5e-3, weight_decay=0.0 if k == 'grid' else 1e-6)
        assert torch.isfinite(loss)
        # the semantic term only exists in batches that drew labelled rays, so the trend is read off rgb + depth
        photometric.append(float(terms['rgb'].detach() + 0.1 * terms['depth'].detach()))
    assert labelled > 0                                   # the class-weighted chunk branch saw the two labelled frames
    assert np.mean(photometric[-4:]) < 0.8 * np.mean(photometric[:4]), photometric
