"""End-to-end parity of the HIP render / loss / backward / Adam path against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as O


def build_pair(encoding='hg+freq', L=16, D=64, C_=3, bound=1.0, seed=0, grid_scale=1e4, log2_T=19, G=15, hidden=128, hidden_color=128):
    """Same parameters in the oracle (named tensors) and in the HIP flat buffer."""
    from autolabel_amd import hip as H
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    cfg = O.ModelConfig(encoding=encoding, feature_dim=D, n_classes=C_, bound=bound, geo_feat_dim=G, hidden_dim=hidden, hidden_dim_color=hidden_color,
                        grid=O.GridSpec(n_levels=L, log2_hashmap_size=log2_T))
    p = O.init_params(cfg, seed)
    if 'grid' in p:
        p['grid'] = (p['grid'] * grid_scale).half().float()  # non-trivial density; exactly representable in the fp16 table
    oracle = O.OracleModel(cfg, params=p, half_sim=True)
    layout = ModelLayout(encoding, G, hidden, hidden_color, D, C_, bound=bound, grid=H.make_grid_desc(n_levels=L, log2_hashmap_size=log2_T))
    P = Params(layout, 'cuda')
    parts = [p['grid'].reshape(-1)] if 'grid' in p else []
    for name in ['sigma', 'color', 'semf', 'semo']:
        parts += [p[f'{name}.{i}'].reshape(-1) for i in range(len(O.mlp_shapes(cfg)[name]))]
    flat = torch.cat([t.detach() for t in parts])
    assert flat.numel() == layout.n_total
    P.flat.copy_(flat.cuda())
    P.refresh_shadows()
    return oracle, HipPipeline(layout, P), cfg


def make_rays(n, seed=0, bound=1.0):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(n, 3, generator=g) - 0.5) * bound
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    norms = 1.0 + torch.rand(n, 1, generator=g) * 0.3
    return o, d, norms


def flat_grads(oracle, cfg):
    parts = [oracle.params['grid'].grad.reshape(-1)] if 'grid' in oracle.params else []
    for name in ['sigma', 'color', 'semf', 'semo']:
        parts += [oracle.params[f'{name}.{i}'].grad.reshape(-1) for i in range(len(O.mlp_shapes(cfg)[name]))]
    return torch.cat(parts)


def rel(a, b):
    return (a - b).norm().item() / max(b.norm().item(), 1e-20)


@pytest.mark.parametrize('S1,S2,perturb', [(128, 128, True), (64, 0, False), (96, 32, False)])
def test_render_forward_matches_oracle(S1, S2, perturb):
    oracle, pipe, cfg = build_pair()
    N = 96
    o, d, norms = make_rays(N, seed=1)
    g = torch.Generator().manual_seed(7)
    noise, u = torch.rand(N, S1, generator=g), torch.rand(N, max(S2, 1), generator=g)
    with torch.no_grad():
        want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=perturb, noise_coarse=noise, u_fine=u)
    od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
    out, ctx = pipe.forward(od, dd, nd, S1, S2, perturb, train=False, noise=nz, u=ud if S2 else None)
    torch.cuda.synchronize()
    # fp16 MLP chain with re-associated fp32 sums + w>1e-4 mask flips: 5e-3 absolute on O(1) outputs
    assert (out['weights_sum'].cpu() - want['weights_sum']).abs().max() < 5e-3
    assert (out['image'].cpu() - want['image']).abs().max() < 5e-3
    assert (out['depth'].cpu() - want['depth']).abs().max() < 5e-3 * max(1.0, want['depth'].max().item())
    assert (out['depth_variance'].cpu() - want['depth_variance']).abs().max() < 1e-2
    assert (out['coordinates_map'].cpu() - want['coordinates_map']).abs().max() < 5e-3
    assert (out['semantic'].cpu() - want['semantic']).abs().max() < 1e-2 * max(1.0, want['semantic'].abs().max().item())
    assert (out['semantic_features'].cpu() - want['semantic_features']).abs().max() < 1e-2 * max(
        1.0, want['semantic_features'].abs().max().item())
    # merged sample order identical to the stable sort of the oracle (index work: exact up to z ties)
    S = S1 + S2
    z_sorted = torch.gather(ctx['z'].cpu().view(-1)[None].expand(N, -1), 1, torch.stack(
        [torch.tensor([r * S1 + i if i < S1 else N * S1 + r * S2 + i - S1 for i in ctx['perm'].cpu()[r].long().tolist()])
         for r in range(N)]))
    assert (z_sorted[:, 1:] >= z_sorted[:, :-1]).all()
    assert (z_sorted - want['_z']).abs().max() < 2e-3


def _batch(N, C_, Cf, seed=0):
    g = torch.Generator().manual_seed(seed)
    b = {'pixels': torch.rand(N, 3, generator=g), 'depth': torch.rand(N, generator=g) * 1.5 + 0.2,
         'semantic': torch.randint(-1, C_, (N,), generator=g), 'features': torch.randn(N, Cf, generator=g)}
    b['depth'][::6] = 0.0
    return b


def hip_loss(pipe, out, batch, N, C_, D, Cf, scale=1.0, weights=(1.0, 0.1, 1.0, 0.5)):
    from autolabel_amd import hip as H
    dv = 'cuda'
    g_image, g_depth = torch.empty(N, 3, device=dv), torch.empty(N, device=dv)
    g_sem, g_feat = torch.empty(N, C_, device=dv), torch.empty(N, D, device=dv)
    counts, terms = torch.zeros(4, dtype=torch.int32, device=dv), torch.zeros(int(H.lib().aln_loss_terms_floats()), device=dv)
    ls = torch.tensor([scale], device=dv)
    gt = {k: v.cuda().contiguous() for k, v in batch.items()}
    gt['semantic'] = gt['semantic'].int()
    H.call('aln_loss_fwd_bwd', H.ptr(out['image']), H.ptr(out['depth']), H.ptr(out['semantic']), H.ptr(out['semantic_features']),
           H.ptr(gt['pixels']), H.ptr(gt['depth']), H.ptr(gt['semantic']), H.ptr(gt['features']) if Cf else None, N, C_, D, Cf,
           weights[0], weights[1], weights[2], weights[3], H.ptr(ls), H.ptr(counts), H.ptr(g_image), H.ptr(g_depth), H.ptr(g_sem),
           H.ptr(g_feat), H.ptr(terms), H.stream())
    torch.cuda.synchronize()  # gt / ls temporaries must outlive the launch
    return g_image, g_depth, g_sem, g_feat, terms


def test_loss_kernel_matches_trainer_formula():
    oracle, pipe, cfg = build_pair(L=2)
    N, C_, D, Cf = 200, 3, 64, 40
    g = torch.Generator().manual_seed(3)
    out = {'image': torch.rand(N, 3, generator=g), 'depth': torch.rand(N, generator=g) * 2,
           'semantic': torch.randn(N, C_, generator=g) * 2, 'semantic_features': torch.randn(N, D, generator=g)}
    batch = _batch(N, C_, Cf)
    o_req = {k: v.clone().requires_grad_(True) for k, v in out.items()}
    loss, terms = O.loss_fn(o_req, batch, feature_loss=True)
    loss.backward()
    out_d = {k: v.cuda() for k, v in out.items()}
    gi, gd, gs, gf, t = hip_loss(pipe, out_d, batch, N, C_, D, Cf, scale=128.0)
    assert abs(t[4].item() - loss.item()) < 1e-5 * max(1, abs(loss.item()))
    assert torch.allclose(gi.cpu() / 128, o_req['image'].grad, atol=1e-7)
    assert torch.allclose(gd.cpu() / 128, o_req['depth'].grad, atol=1e-7)
    assert torch.allclose(gs.cpu() / 128, o_req['semantic'].grad, atol=1e-6)
    assert torch.allclose(gf.cpu() / 128, o_req['semantic_features'].grad, atol=1e-7)
    # no labels / no valid depth: terms vanish instead of NaN (SPEC)
    batch2 = dict(batch, semantic=torch.full((N,), -1), depth=torch.zeros(N))
    gi, gd, gs, gf, t = hip_loss(pipe, out_d, batch2, N, C_, D, Cf)
    assert gs.abs().max().item() == 0 and gd.abs().max().item() == 0 and torch.isfinite(t).all()


@pytest.mark.parametrize('n_classes,G,hidden', [(3, 15, 128), (20, 15, 128), (40, 15, 128), (64, 15, 128), (3, 7, 128), (5, 1, 128), (3, 15, 64)])
def test_train_step_gradients_match_oracle_autograd(n_classes, G, hidden):
    """Backward parity (class counts 3 / 20 / 40 / 64 = logits padded to 16 / 32 / 48 / 64 columns).  The oracle is evaluated at the HIP path's own importance samples (z_fine_override):
    the finest hash-grid cells are 4e-6 wide, so a 1e-6 difference in z (re-associated cumsum in the sampler,
    tested on its own above) moves a sample's gradient to other table entries."""
    oracle, pipe, cfg = build_pair(L=16, D=64, C_=n_classes, G=G, hidden=hidden, hidden_color=hidden)
    assert pipe.recompute == (hidden == 128)   # 64-wide density / color nets: saved activations + generic backward kernels
    N, S1, S2, C_, D, Cf = 64, 64, 64, n_classes, 64, 48
    o, d, norms = make_rays(N, seed=2)
    g = torch.Generator().manual_seed(11)
    noise, u = torch.rand(N, S1, generator=g), torch.rand(N, S2, generator=g)
    batch = _batch(N, C_, Cf, seed=4)
    od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
    out, ctx = pipe.forward(od, dd, nd, S1, S2, True, train=True, noise=nz, u=ud)
    z_fine = ctx['z'][N * S1:].view(N, S2).cpu()
    want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=True, noise_coarse=noise, u_fine=u,
                      z_fine_override=z_fine)
    loss, _ = O.loss_fn(want, batch, feature_loss=True)
    loss.backward()
    gw = flat_grads(oracle, cfg)
    assert (out['image'].cpu() - want['image']).abs().max() < 3e-3
    scale = 1024.0
    gi, gd, gs, gf, t = hip_loss(pipe, out, batch, N, C_, D, Cf, scale=scale)
    assert abs(t[4].item() - loss.item()) < 5e-3 * max(1.0, loss.item())
    pipe.P.grad.zero_()
    pipe.backward(ctx, gi, gd, gs, gf)
    torch.cuda.synchronize()
    assert pipe.found_inf.item() == 0
    got = pipe.P.grad[:pipe.L.n_total].cpu() / scale
    L = pipe.L
    # fp16 gradient activations (loss-scaled) vs fp32 autograd: <= 1% of each tensor's norm
    assert rel(got[:L.n_grid], gw[:L.n_grid]) < 1e-2, 'hash-grid gradient'
    for k in ['sigma', 'color', 'semf', 'semo']:
        a = L.offsets[k]
        b = a + L.nets[k].n_params
        assert rel(got[a:b], gw[a:b]) < 1e-2, k
    touched_h, touched_o = got[:L.n_grid] != 0, gw[:L.n_grid] != 0
    # same table entries receive gradient (fp16 underflow of tiny contributions may drop a few)
    assert (touched_h != touched_o).float().mean().item() < 1e-3


@pytest.mark.parametrize('D,C_', [(512, 40), (64, 100), (128, 7)])
def test_lseg_width_heads_match_oracle(D, C_):
    """LSeg configuration (docs/vision-language.md:19, scripts/ros/node.py:166-176): 512-d feature head, many classes; also
    a 64-d head with 100 classes (only semantic_out is too wide: 112 logit columns) and a 128-d head.
    The wide heads run on the hand-written MFMA GEMMs of wide.hip (no library dispatch).  Forward and gradients vs the oracle."""
    oracle, pipe, cfg = build_pair(L=4, D=D, C_=C_)
    nets = pipe.L.nets
    assert (nets['semf'].wide or nets['semo'].wide) and not nets['sigma'].wide and pipe.L.sem_wide
    assert pipe.L.sem_wide and not hasattr(pipe, '_lib_fwd'), 'the wide heads run on wide.hip: there is no library-GEMM path'
    N, S1, S2, Cf = 24, 32, 32, min(D, 512)
    o, d, norms = make_rays(N, seed=3)
    g = torch.Generator().manual_seed(5)
    noise, u = torch.rand(N, S1, generator=g), torch.rand(N, S2, generator=g)
    batch = _batch(N, C_, Cf, seed=6)
    od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
    out, ctx = pipe.forward(od, dd, nd, S1, S2, True, train=True, noise=nz, u=ud)
    want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=True, noise_coarse=noise, u_fine=u,
                      z_fine_override=ctx['z'][N * S1:].view(N, S2).cpu())
    assert (out['semantic_features'].cpu() - want['semantic_features']).abs().max() < 1e-2 * max(1.0, want['semantic_features'].abs().max().item())
    assert (out['semantic'].cpu() - want['semantic']).abs().max() < 1e-2 * max(1.0, want['semantic'].abs().max().item())
    loss, _ = O.loss_fn(want, batch, feature_loss=True)
    loss.backward()
    gw = flat_grads(oracle, cfg)
    scale = 256.0
    gi, gd, gs, gf, t = hip_loss(pipe, out, batch, N, C_, D, Cf, scale=scale)
    pipe.P.grad.zero_()
    pipe.backward(ctx, gi, gd, gs, gf)
    torch.cuda.synchronize()
    assert pipe.found_inf.item() == 0
    got = pipe.P.grad[:pipe.L.n_total].cpu() / scale
    L = pipe.L
    for k in ['sigma', 'color', 'semf', 'semo']:
        a = L.offsets[k]
        b = a + L.nets[k].n_params
        assert rel(got[a:b], gw[a:b]) < 2e-2, k
    assert rel(got[:L.n_grid], gw[:L.n_grid]) < 2e-2


def test_lseg_linear_last_layer_per_ray_matches_the_per_sample_oracle():
    """semantic_weight = 0 (scripts/ros/node.py:166-176): the wide heads skip semantic_out and apply the last, linear layer of
    semantic_features once per ray to the composited hidden activation.  Outputs and every gradient against the oracle, which
    evaluates the heads per sample (autolabel/models.py:248-256) under the same loss weights."""
    D, C_ = 512, 5
    oracle, pipe, cfg = build_pair(L=4, D=D, C_=C_)
    N, S1, S2, Cf = 24, 32, 32, 512
    o, d, norms = make_rays(N, seed=3)
    g = torch.Generator().manual_seed(5)
    noise, u = torch.rand(N, S1, generator=g), torch.rand(N, S2, generator=g)
    batch = _batch(N, C_, Cf, seed=6)
    od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
    out, ctx = pipe.forward(od, dd, nd, S1, S2, True, train=True, noise=nz, u=ud, sem_linear=True)
    assert ctx['sem_linear'] and out['semantic'].abs().max().item() == 0
    want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=True, noise_coarse=noise, u_fine=u,
                      z_fine_override=ctx['z'][N * S1:].view(N, S2).cpu())
    assert (out['semantic_features'].cpu() - want['semantic_features']).abs().max() < 1e-2 * max(1.0, want['semantic_features'].abs().max().item())
    assert (out['image'].cpu() - want['image']).abs().max() < 5e-3
    loss, _ = O.loss_fn(want, batch, feature_loss=True, semantic_weight=0.0)
    loss.backward()
    gw = flat_grads(oracle, cfg)
    scale = 256.0
    gi, gd, gs, gf, t = hip_loss(pipe, out, batch, N, C_, D, Cf, scale=scale, weights=(1.0, 0.1, 0.0, 0.5))
    assert gs.abs().max().item() == 0
    pipe.P.grad.zero_()
    pipe.backward(ctx, gi, gd, gs, gf)
    torch.cuda.synchronize()
    assert pipe.found_inf.item() == 0
    L = pipe.L
    got = pipe.P.grad[:L.n_total].cpu() / scale
    assert rel(got[:L.n_grid], gw[:L.n_grid]) < 2e-2, 'hash-grid gradient'
    for k in ['sigma', 'color', 'semf']:
        a = L.offsets[k]
        b = a + L.nets[k].n_params
        assert rel(got[a:b], gw[a:b]) < 2e-2, k
    a = L.offsets['semo']
    assert got[a:a + L.nets['semo'].n_params].abs().max().item() == 0 and gw[a:a + L.nets['semo'].n_params].abs().max().item() == 0


def test_training_converges_like_the_oracle():
    """Matched quality: the HIP engine and the fp32 CPU oracle train the same small model (S0 cube scene, L=4, T=2^14) from
    the same initialisation on the same batches / random numbers for 50 Adam steps; the loss trajectories and the PSNR of
    a training view agree (fp16 vs fp32 arithmetic, independently drawn importance samples)."""
    import math
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import ArrayDataset
    from autolabel_amd.engine import TrainEngine
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle, pipe, cfg = build_pair(L=4, D=64, C_=3, bound=6.0, grid_scale=1.0, log2_T=14)
    oracle.half_sim = False
    scene = synthetic.make_cube_scene()
    ds = ArrayDataset(scene, batch_size=512)
    import random
    np.random.seed(0); random.seed(0)
    N, S1, S2, steps = 512, 32, 32, 50
    eng = TrainEngine(pipe, num_steps=S1, upsample_steps=S2, feature_loss=False)
    st = {k: [torch.zeros_like(v), torch.zeros_like(v), 0] for k, v in oracle.params.items()}
    lh, lo = [], []
    for it in range(steps):
        b = ds._next_train()
        bt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
        dev = {k: v.cuda().float().contiguous() for k, v in bt.items() if k != 'semantic'}
        dev['semantic'] = bt['semantic'].int().cuda()
        eng.step(dev, seed=7, step=it)
        lh.append(eng.terms[4].item())
        noise = torch.from_numpy(O.rand_uniform(7, O.STREAM_PERTURB, it, np.arange(N * S1))).view(N, S1)
        u = torch.from_numpy(O.rand_uniform(7, O.STREAM_PDF, it, np.arange(N * S2))).view(N, S2)
        out = oracle.run(bt['rays_o'], bt['rays_d'], bt['direction_norms'], S1, S2, perturb=True, noise_coarse=noise, u_fine=u)
        loss, _ = O.loss_fn(out, {'pixels': bt['pixels'], 'depth': bt['depth'], 'semantic': bt['semantic']})
        for p in oracle.params.values():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for k, p in oracle.params.items():
                if p.grad is None:   # torch.optim.Adam skips tensors without gradient (heads unused by this batch's loss)
                    continue
                st[k][2] += 1
                O.adam_update(p, p.grad, st[k][0], st[k][1], st[k][2], 5e-3, weight_decay=0.0 if k == 'grid' else 1e-6)
        lo.append(loss.item())
    assert eng.state_i[0].item() == steps, 'no step may be skipped by the loss scaler in this run'
    assert lo[-1] < 0.6 * lo[0] and lh[-1] < 0.6 * lh[0]
    # same trajectory: mean loss over the last 10 steps within 10 %
    a, b_ = sum(lh[-10:]) / 10, sum(lo[-10:]) / 10
    assert abs(a - b_) < 0.1 * b_, (a, b_)
    t = ds._get_test(0)
    ro, rd, dn = [torch.from_numpy(np.ascontiguousarray(t[k])).float() for k in ['rays_o', 'rays_d', 'direction_norms']]
    gt = torch.from_numpy(t['pixels']).reshape(-1, 3)
    with torch.no_grad():
        img_o = oracle.run(ro.reshape(-1, 3), rd.reshape(-1, 3), dn, 64, 0, perturb=False)['image']
    img_h, _ = pipe.forward(ro.reshape(-1, 3).cuda(), rd.reshape(-1, 3).cuda(), dn.reshape(-1).cuda(), 64, 0, False, train=False)
    psnr = lambda x: -10 * math.log10(((x - gt) ** 2).mean().item())
    ph, po = psnr(img_h['image'].cpu()), psnr(img_o)
    assert abs(ph - po) < 1.0, (ph, po)   # +-1 dB after 50 steps from identical init


def test_adam_step_reads_the_fp16_payload_of_the_exchange_bit_for_bit():
    """Data parallelism, fp16 on the wire: `aln_adam_step_wire` takes the table's averaged gradient from the halves the exchange left
    (+ a watch-only `aln_grad_unpack_f16`) -- the same step, bit for bit, as unpacking them into the fp32 gradient and running
    `aln_adam_step`; a non-finite half is seen by the watch and the step skipped; the fp32 table gradient is neither read nor cleared."""
    from autolabel_amd import hip as H
    import ctypes
    n_grid, n = 4096, 4096 + 600
    g = torch.Generator().manual_seed(4)
    ends, kinds = (ctypes.c_int64 * 2)(n_grid, n), (ctypes.c_int32 * 2)(0, 0)
    state = lambda: (torch.zeros(16, dtype=torch.int32, device='cuda'), torch.tensor([128.0, 0, 0, 0], device='cuda'), torch.zeros(24, device='cuda'))
    p0 = torch.randn(n, generator=g).cuda()
    A = dict(p=p0.clone(), gr=torch.zeros(n).cuda(), m=torch.zeros(n).cuda(), v=torch.zeros(n).cuda(), t16=torch.zeros(n_grid, dtype=torch.float16, device='cuda'))
    B = {k: t.clone() for k, t in A.items()}
    (siA, sfA, cA), (siB, sfB, cB) = state(), state()
    sentinel = 123.0
    for it in range(5):
        wire = (torch.randn(n_grid, generator=g) * (10.0 ** float(torch.randint(-6, 3, (1,), generator=g)))).half().cuda()
        if it == 3:
            wire[777] = float('inf')          # an overflow of the fp16 SUM: every rank sees it in the reduced halves
        mlp = torch.randn(n - n_grid, generator=g).cuda()
        # reference route: unpack into the fp32 gradient, then the plain step
        A['gr'][n_grid:] = mlp
        H.call('aln_grad_unpack_f16', H.ptr(wire), n_grid, H.ptr(A['gr']), H.ptr(siA[2:3]), H.stream())
        H.call('aln_adam_step', H.ptr(A['p']), H.ptr(A['gr']), H.ptr(A['m']), H.ptr(A['v']), H.ptr(A['t16']), n_grid, n, H.ptr(siA), H.ptr(sfA), H.ptr(cA),
               5e-3, 0.9, 0.99, 1e-15, 1e-6, 2.0, 0.5, 3, 2, ends, kinds, 0, 0, None, None, H.stream())
        # wire route: watch only, the optimizer reads the halves; the fp32 table gradient holds a sentinel nobody may touch
        B['gr'][:n_grid] = sentinel
        B['gr'][n_grid:] = mlp
        H.call('aln_grad_unpack_f16', H.ptr(wire), n_grid, None, H.ptr(siB[2:3]), H.stream())
        H.call('aln_adam_step_wire', H.ptr(B['p']), H.ptr(B['gr']), H.ptr(B['m']), H.ptr(B['v']), H.ptr(B['t16']), n_grid, n, H.ptr(siB), H.ptr(sfB), H.ptr(cB),
               5e-3, 0.9, 0.99, 1e-15, 1e-6, 2.0, 0.5, 3, 2, ends, kinds, 0, H.ptr(wire), None, None, H.stream())
        torch.cuda.synchronize()
        for k in ('p', 'm', 'v', 't16'):
            assert torch.equal(A[k].view(torch.int32 if k != 't16' else torch.int16), B[k].view(torch.int32 if k != 't16' else torch.int16)), (it, k)
        assert torch.equal(siA, siB) and torch.equal(sfA, sfB), it
        assert bool((B['gr'][:n_grid] == sentinel).all()) and float(B['gr'][n_grid:].abs().max()) == 0.0
    assert int(siA[0]) == 4 and float(sfA[0]) == 128.0 * 2.0 * 0.5      # grew after three clean steps, the overflow step was skipped and backed off
    with pytest.raises(RuntimeError):       # the halves can only stand in for the whole table under the replicated optimizer
        H.call('aln_adam_step_wire', H.ptr(B['p']), H.ptr(B['gr']), H.ptr(B['m']), H.ptr(B['v']), H.ptr(B['t16']), n_grid, n, H.ptr(siB), H.ptr(sfB), H.ptr(cB),
               5e-3, 0.9, 0.99, 1e-15, 1e-6, 2.0, 0.5, 3, 2, ends, kinds, 0, None, None, None, H.stream())


def test_adam_step_matches_torch_adam_and_skips_on_inf():
    from autolabel_amd import hip as H
    n_grid, n = 1000, 1600
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{'params': [p_ref]}], lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    p, gr, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    t16 = torch.zeros(n_grid, dtype=torch.float16, device='cuda')
    si, sf, cst = torch.zeros(16, dtype=torch.int32, device='cuda'), torch.tensor([128.0, 0, 0, 0], device='cuda'), torch.zeros(24, device='cuda')
    import ctypes
    ends, kinds = (ctypes.c_int64 * 2)(n_grid, n), (ctypes.c_int32 * 2)(0, 1)
    counts = torch.tensor([0, 1], dtype=torch.int32, device='cuda')

    def step():
        H.call('aln_adam_step', H.ptr(p), H.ptr(gr), H.ptr(m), H.ptr(v), H.ptr(t16), n_grid, n, H.ptr(si), H.ptr(sf), H.ptr(cst),
               5e-3, 0.9, 0.99, 1e-15, 1e-6, 2.0, 0.5, 3, 2, ends, kinds, 0, 0, H.ptr(counts), None, H.stream())
    for it in range(4):
        gt = torch.randn(n, generator=g)
        pr = p_ref.detach().clone()
        p_ref.grad = gt.clone()
        p_ref.grad[n_grid:] += 1e-6 * pr[n_grid:]   # weight decay on the MLP group only (scripts/train.py:55-58)
        opt.step()
        scale = sf[0].item()
        gr.copy_((gt * scale).cuda())
        step()
        assert gr.abs().max().item() == 0  # gradients are zeroed by the fused step
    assert torch.allclose(p.cpu(), p_ref.detach(), atol=2e-6)
    assert torch.equal(t16.cpu(), p[:n_grid].cpu().half())
    assert si[0].item() == 4 and sf[0].item() == 256.0  # grew once after 3 clean steps
    # an overflow step is skipped, scale backs off, step counter is unchanged
    before = p.clone()
    gr.fill_(1.0); si[2] = 1
    step()
    assert torch.equal(p, before) and sf[0].item() == 128.0 and si[0].item() == 4 and si[2].item() == 0
    # a block whose gradient is None in the reference (no labelled ray) is left untouched, its step counter too
    counts[1] = 0
    gr.fill_(128.0)
    step()
    assert torch.equal(p[n_grid:], before[n_grid:]) and not torch.equal(p[:n_grid], before[:n_grid])
    assert si[4].item() == 5 and si[5].item() == 4


def test_inference_and_training_forward_agree_bitwise():
    """train=False skips the color_in tensor (inputs built inside the color kernel): same bits -- with the semantic outputs as
    per-tile sums (sample counts multiples of 32: no f / logits rows in either mode) and as composited rows (other counts); the
    two forms of the same sums agree to fp32 rounding."""
    _, pipe, _ = build_pair(C_=5)
    o, d, norms = make_rays(600, seed=4)
    od, dd, nd = o.cuda(), d.cuda(), norms.reshape(-1).cuda()
    got = {}
    for S1, S2, sums in [(64, 32, True), (64, 24, False)]:
        a, ctx = pipe.forward(od, dd, nd, S1, S2, False, train=True)
        assert bool(ctx.get('sem_sums')) == sums
        a = {k: v.clone() for k, v in a.items()}
        b, ctx_b = pipe.forward(od, dd, nd, S1, S2, False, train=False)
        assert bool(ctx_b.get('sem_sums')) == sums
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # the row path against the sums path on the same samples: force rows by handing the fused pair a sample count it refuses
    a, _ = pipe.forward(od, dd, nd, 64, 32, False, train=False)
    a = {k: v.clone() for k, v in a.items()}
    from autolabel_amd import hip as hip_mod
    lib = hip_mod.lib()
    saved = lib.aln_sem_heads_bwd_slabs
    try:
        lib.aln_sem_heads_bwd_slabs = lambda *args: 0
        b, ctx_b = pipe.forward(od, dd, nd, 64, 32, False, train=False)
    finally:
        lib.aln_sem_heads_bwd_slabs = saved
    assert not ctx_b.get('sem_sums')
    for k in ('semantic', 'semantic_features'):
        assert (a[k] - b[k]).abs().max().item() <= 2e-6 * b[k].abs().max().item() + 1e-7, k
    assert torch.equal(a['image'], b['image'])


def test_diverged_field_does_not_corrupt_memory():
    """NaN / inf parameters (a diverged run) must give NaN outputs, not out-of-range indices: the sample merge orders NaN
    distances like +inf, positions are clamped, so every index stays inside its buffer (canary rows around the workspaces)."""
    oracle, pipe, cfg = build_pair(L=16, D=64, C_=3)
    with torch.no_grad():
        pipe.P.flat[pipe.L.n_grid:pipe.L.n_grid + 2000] = float('nan')
        pipe.P.flat[pipe.L.n_grid + 2000:pipe.L.n_grid + 4000] = float('inf')
    pipe.P.refresh_shadows()
    N, S1, S2 = 256, 64, 64
    o, d, norms = make_rays(N, seed=5)
    canary = torch.full((1 << 20,), 7.0, device='cuda')      # allocated after the workspaces of earlier tests, before this pipe's
    out, ctx = pipe.forward(o.cuda(), d.cuda(), norms.cuda().reshape(-1), S1, S2, True, train=True, seed=1, step=0)
    gi = torch.ones(N, 3, device='cuda'); gd = torch.ones(N, device='cuda')
    gs = torch.zeros(N, 3, device='cuda'); gf = torch.zeros(N, 64, device='cuda')
    pipe.P.grad.zero_()
    pipe.backward(ctx, gi, gd, gs, gf)
    torch.cuda.synchronize()
    perm = ctx['perm'].view(N, S1 + S2).long().cpu()
    assert ((perm >= 0) & (perm < S1 + S2)).all() and (perm.sort(dim=1).values == torch.arange(S1 + S2)[None]).all(), 'merge must stay a permutation'
    assert (canary == 7.0).all()
    assert pipe.found_inf.item() == 1
