"""Memory discipline of the launch sequence ("a kernel reading scratch it did not write" / a stray write would be a
latent bug of the single-GPU product too).  Every workspace buffer is carved out of a larger allocation with canary bands on either
side and starts as NaN bytes instead of whatever the allocator returned (`pipeline.Workspace.guard_bytes`); the same steps then have
to (a) leave every canary intact and (b) reproduce the unguarded run BIT FOR BIT -- a result that depended on stale scratch or on the
placement of the buffers could not.  Plus the regression for the fault this round's stress run found: a batch without a single live
sample (`n_live == 0`) sent the colour head's prefetch 4 KB in front of its input buffer."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(D=64, cuda_ray=False, classes=None):
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
    if D == 512:
        scene = dict(scene)
        g = torch.Generator().manual_seed(1)
        scene['features'] = torch.randn(8, 4 * 4, 64, generator=g).half().to(dev)
        scene['feat_hw'] = (4, 4)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, D, classes or scene['n_classes'], bound=3.0)
    P = Params(layout, dev); P.init_(seed=0)
    with torch.no_grad():
        P.flat[:layout.n_grid].mul_(2e3)       # a table that shapes the density: live and dead rows for the colour head
    P.refresh_shadows()
    pipe = HipPipeline(layout, P)
    if cuda_ray:
        pipe.enable_marching(G=64, max_steps=256, samples=32)
        pipe.occ.bits.fill_(-1)
    return frames, layout, P, pipe


def _run(kind, guard):
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import Workspace
    Workspace.guard_bytes = 4096 if guard else 0
    try:
        frames, layout, P, pipe = _setup(D=512 if kind == 'lseg' else 64, cuda_ray=kind == 'march')
        pipe.tiled_enc_train = kind == 'groups'      # (one of the four kinds trains through the tiled hash-grid output as well)
        pipe.phased_min_rows = 1 << 15               # (... at this test's 32 K rows per pass)
        kw = dict(feature_loss=True, semantic_weight=0.0) if kind == 'lseg' else {}
        eng = TrainEngine(pipe, num_steps=32, upsample_steps=32, **kw)
        batch = frames.alloc_batch(512 if kind == 'lseg' else 1024)
        groups = eng.level_groups()
        for i in range(3):
            frames.next_train(batch, seed=5, step=i)
            if kind == 'groups':    # the data-parallel launch sequence (scatter in level groups, separate optimizer pass), no collectives
                eng.fuse_grid_adam = False
                eng.P.grad.zero_()
                eng.maybe_update_grid()
                out, ctx = pipe.forward(batch['rays_o'], batch['rays_d'], batch['direction_norms'].reshape(-1), eng.S1, eng.S2, True, train=True,
                                        seed=7, step=i, ws=eng.ws)
                g = [torch.full_like(out[k], 1e-3) for k in ('image', 'depth', 'semantic', 'semantic_features')]
                pipe.backward(ctx, g[0], g[1], g[2], g[3], level_groups=groups, on_grad_ready=lambda *a: None, scatter_flag=eng.state_i[3:4])
                eng.optimizer_step()
            else:
                eng.step(batch, seed=7, step=i)
        t = frames.get_test(1)
        with torch.no_grad():
            r, _ = pipe.forward(t['rays_o'].reshape(-1, 3), t['rays_d'].reshape(-1, 3), t['direction_norms'].reshape(-1), 64, 0 if kind == 'march' else 32,
                                False, train=False, march=kind == 'march')
        torch.cuda.synchronize()
        bad = eng.ws.check_guards() + pipe.ws.check_guards()
        res = dict(flat=P.flat.clone(), m=eng.m.clone(), v=eng.v.clone(), steps=int(eng.state_i[0]),
                   render={k: v.clone() for k, v in r.items()}, n_guarded=len(eng.ws._guarded) + len(pipe.ws._guarded))
        return res, bad
    finally:
        Workspace.guard_bytes = 0


@pytest.mark.parametrize('kind', ['dense', 'groups', 'march', 'lseg'])
def test_guarded_poisoned_workspaces_change_nothing(kind):
    ref, _ = _run(kind, guard=False)
    got, bad = _run(kind, guard=True)
    assert got['n_guarded'] > 20 and ref['n_guarded'] == 0
    assert not bad, f'canary bands overwritten (buffer, bytes below, bytes above): {bad}'
    assert ref['steps'] == got['steps'] == 3
    for k in ('flat', 'm', 'v'):
        assert torch.isfinite(got[k]).all()
        assert torch.equal(ref[k], got[k]), f'{kind}: {k} differs in {(ref[k] != got[k]).sum().item()} elements between plain and guarded / NaN-filled workspaces'
    for k, v in ref['render'].items():
        assert torch.equal(v, got['render'][k]), (kind, 'render', k)


def test_a_batch_without_a_live_sample_steps_safely():
    """Rays that miss the box: every compositing weight is 0, so no row reaches the colour head (`n_live == 0`, models.py:199-200 returns
    zeros for an empty mask).  k_mlp_fwd128's prefetch used to clamp its row index to rows - 1 = -1 and read in front of the buffer."""
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import Workspace
    Workspace.guard_bytes = 4096
    try:
        frames, layout, P, pipe = _setup()
        eng = TrainEngine(pipe, num_steps=32, upsample_steps=32)
        batch = frames.alloc_batch(1024)
        frames.next_train(batch, seed=5, step=0)
        batch['rays_o'][:] = torch.tensor([50.0, 50.0, 50.0], device='cuda')        # far outside [-3, 3]^3 ...
        batch['rays_d'][:] = torch.nn.functional.normalize(torch.tensor([[1.0, 0.5, 0.25]], device='cuda'))   # ... looking away
        before = P.flat.clone()
        for i in range(2):
            out = eng.step(batch, seed=7, step=i)
        torch.cuda.synchronize()
        assert int(eng.ws.bufs['n_live'][1].item()) == 0
        assert not eng.ws.check_guards()
        assert torch.isfinite(P.flat).all() and all(torch.isfinite(v).all() for v in out.values())
        assert float(out['weights_sum'].abs().max()) == 0.0 and bool((out['image'] == 1.0).all()), 'an empty ray shows the white background'
        assert int(eng.state_i[0]) == 2 and not torch.equal(before, P.flat)
    finally:
        Workspace.guard_bytes = 0
