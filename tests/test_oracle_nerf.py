"""Known-answer and self-consistency tests of oracle/nerf_oracle.py (parity unpinned:
no reference vectors exist for this half -- see oracle/__init__.py)."""
import math

import numpy as np
import torch

from oracle import nerf_oracle as O


def test_grid_level_table_matches_reference_config():
    # autolabel/models.py:38-48 -> 14,229,504 parameters
    spec = O.GridSpec()
    lv = spec.levels()
    assert [l['res'] for l in lv[:4]] == [16, 32, 64, 128]
    assert [l['size'] for l in lv[:4]] == [4096, 32768, 262144, 524288]
    assert [l['dense'] for l in lv[:4]] == [True, True, True, False]
    assert lv[15]['scale'] == 16 * 2 ** 15 - 1
    assert spec.n_entries * spec.n_features == 14229504


def test_grid_index_known_answers():
    """Hand arithmetic with Python ints (independent of the torch code path)."""
    spec = O.GridSpec()
    lv = spec.levels()
    x = torch.tensor([[0.5, 0.5, 0.5], [0.25, 0.5, 0.75], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]], dtype=torch.float32)
    idx0, w0 = O.grid_corner_indices(x, lv[0])
    # level 0: scale 15 -> pos 8.0 -> g=8, frac 0 -> corner 0 carries all weight
    assert idx0[0, 0].item() == 8 + 8 * 16 + 8 * 256 and w0[0, 0].item() == 1.0
    assert idx0[0, 7].item() == 9 + 9 * 16 + 9 * 256
    # x=1: pos 15.5 -> g=15, corner 7 -> (16,16,16) wraps modulo 4096
    assert idx0[2, 7].item() == (16 + 16 * 16 + 16 * 256) % 4096
    assert idx0[3, 0].item() == 0 and abs(w0[3, 0].item() - 0.125) < 1e-7
    # level 3: scale 127, hashed, T = 2^19
    idx3, w3 = O.grid_corner_indices(x, lv[3])
    gx, gy, gz = 32, 64, 95  # floor(0.25*127+.5), floor(0.5*127+.5), floor(0.75*127+.5)
    want = (gx ^ ((gy * 2654435761) & 0xFFFFFFFF) ^ ((gz * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert idx3[1, 0].item() == want
    want7 = ((gx + 1) ^ (((gy + 1) * 2654435761) & 0xFFFFFFFF) ^ (((gz + 1) * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert idx3[1, 7].item() == want7
    assert torch.allclose(w3.sum(1), torch.ones(4), atol=1e-6)
    # corner bit order: bit0 = x
    assert idx3[1, 1].item() == ((gx + 1) ^ ((gy * 2654435761) & 0xFFFFFFFF) ^ ((gz * 805459861) & 0xFFFFFFFF)) % (1 << 19)


def test_hashgrid_constant_table_interpolates_to_constant():
    spec = O.GridSpec(n_levels=4)
    table = torch.full((spec.n_entries, 2), 0.25)
    x = torch.rand(100, 3)
    out = O.hashgrid_encode(x, table, spec)
    assert out.shape == (100, 8) and torch.allclose(out, torch.full_like(out, 0.25), atol=1e-6)


def test_freq_and_sh_known_values():
    x = torch.tensor([[0.25, -0.5, 1.0]])
    f = O.freq_encode(x, 2)
    assert f.shape == (1, 12)
    want = [math.sin(math.pi * .25), math.cos(math.pi * .25), math.sin(2 * math.pi * .25), math.cos(2 * math.pi * .25)]
    assert np.allclose(f[0, :4].numpy(), want, atol=1e-6)
    assert abs(f[0, 4].item() - math.sin(-math.pi * .5)) < 1e-6
    sh = O.sh4_encode(torch.tensor([[0.5, 0.5, 1.0]]))  # d = (0,0,1)
    assert abs(sh[0, 0].item() - 0.28209479) < 1e-7 and abs(sh[0, 2].item() - 0.48860251) < 1e-7
    assert abs(sh[0, 6].item() - (0.94617470 - 0.31539157)) < 1e-6
    assert abs(sh[0, 12].item() - 0.37317633 * 2.0) < 1e-6
    assert sh[0, [1, 3, 4, 5, 7, 8, 9, 10, 11, 13, 14, 15]].abs().max().item() < 1e-7


def test_mlp_pads_input_with_ones():
    W = [torch.zeros(16, 16), torch.eye(16)]
    W[0][0, 15] = 2.0  # only the padded column is connected
    y = O.mlp_forward(torch.zeros(3, 15), W)
    assert torch.allclose(y[:, 0], torch.full((3,), 2.0)) and y[:, 1:].abs().max() == 0


def test_param_count_matches_survey():
    cfg = O.ModelConfig()
    n = sum(o * i for shapes in O.mlp_shapes(cfg).values() for o, i in shapes)
    assert n == 24576 + 22528 + 9216 + 6144
    assert cfg.enc_dim == 44


def test_adam_matches_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(50)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [p_ref], 'weight_decay': 1e-6}], lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    p, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for t in range(1, 6):
        g = torch.randn(50)
        p_ref.grad = g.clone()
        opt.step()
        O.adam_update(p, g, m, v, t, 5e-3, weight_decay=1e-6)
    assert torch.allclose(p, p_ref.detach(), atol=1e-7)


def test_loss_matches_trainer_formula():
    """Literal autolabel/trainer.py:72-92 written with torch.nn.functional."""
    torch.manual_seed(1)
    B, C, D = 64, 5, 8
    out = {'image': torch.rand(B, 3), 'depth': torch.rand(B) * 3, 'semantic': torch.randn(B, C),
           'semantic_features': torch.randn(B, D)}
    batch = {'pixels': torch.rand(B, 3), 'depth': torch.rand(B) * 3, 'semantic': torch.randint(-1, C, (B,)),
             'features': torch.randn(B, 6)}
    batch['depth'][::5] = 0
    loss, _ = O.loss_fn(out, batch, feature_loss=True)
    F = torch.nn.functional
    want = 1.0 * torch.nn.MSELoss(reduction='none')(out['image'], batch['pixels']).mean()
    hd = batch['depth'] > 0.01
    want = want + 0.1 * torch.abs(out['depth'][hd] - batch['depth'][hd]).mean()
    want = want + 0.5 * F.l1_loss(out['semantic_features'][:, :6], batch['features'])
    hs = batch['semantic'] >= 0
    want = want + 1.0 * F.cross_entropy(out['semantic'][hs, :], batch['semantic'][hs])
    assert torch.allclose(loss, want)


def _small_model(half_sim=False):
    cfg = O.ModelConfig(grid=O.GridSpec(n_levels=4), feature_dim=64, n_classes=3, bound=1.0)
    cfg.enc_dim  # 12 + 8 = 20 -> padded 32
    return O.OracleModel(cfg, half_sim=half_sim, seed=0)


def _rays(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(n, 3, generator=g) - 0.5) * 0.5
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    return o, d, torch.ones(n, 1) * 1.1


def test_render_shapes_and_gradients():
    m = _small_model()
    # bigger grid values so density is non-trivial
    with torch.no_grad():
        m.params['grid'].mul_(1e4)
    o, d, nrm = _rays(32)
    g = torch.Generator().manual_seed(3)
    out = m.run(o, d, nrm, num_steps=16, upsample_steps=16, perturb=True,
                noise_coarse=torch.rand(32, 16, generator=g), u_fine=torch.rand(32, 16, generator=g))
    assert out['image'].shape == (32, 3) and out['semantic'].shape == (32, 3)
    assert out['semantic_features'].shape == (32, 64) and out['coordinates_map'].shape == (32, 3)
    assert (out['_z'][:, 1:] >= out['_z'][:, :-1]).all()
    assert (out['weights_sum'] <= 1 + 1e-5).all() and (out['depth'] >= 0).all()
    batch = {'pixels': torch.rand(32, 3), 'depth': torch.rand(32) + 0.5, 'semantic': torch.randint(-1, 3, (32,)),
             'features': torch.randn(32, 64)}
    loss, terms = O.loss_fn(out, batch, feature_loss=True)
    loss.backward()
    for k, v in m.params.items():
        assert v.grad is not None and torch.isfinite(v.grad).all(), k
        assert v.grad.abs().sum() > 0, k


def test_render_deterministic_path_and_half_sim_close():
    o, d, nrm = _rays(16, seed=5)
    outs = []
    for hs in (False, True):
        m = _small_model(hs)
        with torch.no_grad():
            m.params['grid'].mul_(1e4)
        outs.append(m.run(o, d, nrm, num_steps=32, upsample_steps=0, perturb=False))
    assert torch.allclose(outs[0]['image'], outs[1]['image'], atol=2e-2)
    assert torch.allclose(outs[0]['depth'], outs[1]['depth'], atol=2e-2)


def test_sample_pdf_deterministic_sorted_inside_bins():
    bins = torch.linspace(1, 2, 9)[None]
    w = torch.rand(1, 8)
    u = (torch.arange(32, dtype=torch.float32) + 0.5) / 32
    z = O.sample_pdf(bins, w, u[None])
    assert (z[:, 1:] >= z[:, :-1]).all() and z.min() >= 1 and z.max() <= 2


def test_near_far_miss_and_inside():
    m = _small_model()
    o = torch.tensor([[0., 0., 0.], [5., 5., 5.], [0., 0., -3.]])
    d = torch.tensor([[0., 0., 1.], [1., 0., 0.], [0., 0., 1.]])
    near, far = m.near_far(o, d)
    mn = np.float32(0.2).item()
    assert near[0].item() == mn and far[0].item() == 1.0
    assert near[1].item() == far[1].item() == mn  # miss
    assert near[2].item() == 2.0 and far[2].item() == 4.0


def test_rng_reference_values_are_stable():
    r = O.rand_u32(1234, 2, 7, np.arange(4))
    assert r.dtype == np.uint32 and len(set(r.tolist())) == 4
    u = O.rand_uniform(1234, 2, 7, np.arange(10000))
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02


def test_fused_grid_position_differs_only_where_the_two_roundings_do():
    """GridSpec.pos_fma (tcnn's fused x * scale + 0.5): evaluated through fp64, where product and sum are exact, so the single
    rounding is the fma result.  Known answer: x = 1/3 (0x3EAAAAAB), scale = 524287: unfused fl(fl(x*s) + 0.5) vs fused fl(x*s + 0.5)."""
    x, s = np.float32(1.0 / 3.0), np.float32(524287.0)
    unfused = np.float32(np.float32(x * s) + np.float32(0.5))
    fused = np.float32(np.float64(x) * np.float64(s) + 0.5)
    spec_u, spec_f = O.GridSpec(), O.GridSpec(pos_fma=True)
    xn = torch.tensor([[float(x)] * 3], dtype=torch.float32)
    for spec, want in ((spec_u, unfused), (spec_f, fused)):
        lv = spec.levels()[15]
        assert lv['scale'] == 524287.0
        idx, w = O.grid_corner_indices(xn, lv)
        frac = float(want - np.floor(want))
        assert abs(float(w[0, 7]) - frac ** 3) < 1e-6       # corner (1,1,1) weight = frac^3
    # on random points the two specs agree except for a small share of last-bit cases
    g = torch.Generator().manual_seed(0)
    xs = torch.rand(20000, 3, generator=g)
    iu, wu = O.grid_corner_indices(xs, spec_u.levels()[15])
    if_, wf = O.grid_corner_indices(xs, spec_f.levels()[15])
    assert 0 < (wu != wf).any(dim=1).float().mean().item() < 0.9
    assert (iu != if_).any(dim=1).float().mean().item() < 1e-3          # a different CELL only when pos rounds across an integer
