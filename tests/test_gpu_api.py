"""The drop-in Python surface (models / renderer / trainer / scripts) on the GPU, checked against the oracle."""
import math
import os
import subprocess
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_model(D=64, C_=3, bound=1.0, encoding='hg+freq', grid_scale=3e3):
    from autolabel_amd.models import ALNetwork
    m = ALNetwork(encoding=encoding, num_layers=2, hidden_dim=128, geo_feat_dim=15, num_layers_color=2, hidden_dim_color=128,
                  hidden_dim_semantic=D, semantic_classes=C_, bound=bound, cuda_ray=False, density_scale=1)
    with torch.no_grad():
        if m._layout.n_grid:
            m.encoder.grid_encoding.params.mul_(grid_scale)
    return m.cuda()


def oracle_of(model):
    """Oracle with the model's current parameters (half_sim: same fp16 rounding points)."""
    L = model._layout
    cfg = O.ModelConfig(encoding=model.encoding, feature_dim=L.D, n_classes=L.C, bound=float(model.bound),
                        grid=O.GridSpec(per_level_scale=float(L.enc.grid.per_level_scale)))
    p = {}
    if L.n_grid:
        p['grid'] = model.encoder.grid_encoding.params.detach().cpu().view(-1, 2).half().float().clone()
    for name, blk in [('sigma', model.sigma_net), ('color', model.color_net), ('semf', model.semantic_features), ('semo', model.semantic_out)]:
        flat, o = blk.params.detach().cpu(), 0
        for i, (no, ni) in enumerate(O.mlp_shapes(cfg)[name]):
            p[f'{name}.{i}'] = flat[o:o + no * ni].view(no, ni).clone()
            o += no * ni
    return O.OracleModel(cfg, params=p, half_sim=True), cfg


@pytest.mark.parametrize('encoding', ['hg+freq', 'freq', 'hg'])
def test_point_queries_match_oracle(encoding):
    model = make_model(encoding=encoding, bound=1.5)
    oracle, cfg = oracle_of(model)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(500, 3, generator=g) * 2 - 1) * 1.5
    d = torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=1)
    with torch.no_grad():
        want = oracle.density(x)
        got = model.density(x.cuda())
        assert got['geo_feat'].shape == (500, 15)
        assert (got['geo_feat'].float().cpu() - want['geo_feat']).abs().max() < 5e-3 * max(1.0, want['geo_feat'].abs().max().item())
        assert (got['sigma'].cpu() / want['sigma'] - 1).abs().max() < 2e-2  # exp() of a 1-ulp-different fp16 logit
        mask = torch.rand(500, generator=g) > 0.5
        rgb_w = oracle.color(x, d, mask=mask, geo_feat=want['geo_feat'])
        rgb = model.color(x.cuda(), d.cuda(), mask=mask.cuda(), geo_feat=want['geo_feat'].cuda())
        assert (rgb.cpu() - rgb_w).abs().max() < 3e-3 and rgb.cpu()[~mask].abs().max() == 0
        assert model.color(x.cuda(), d.cuda(), mask=torch.zeros(500, dtype=torch.bool).cuda(), geo_feat=want['geo_feat'].cuda()).abs().max() == 0
        lw, fw = oracle.semantic(want['geo_feat'])
        lg, fg = model.semantic(want['geo_feat'].cuda(), got['sigma'])
        assert (lg.float().cpu() - lw).abs().max() < 5e-3 * max(1.0, lw.abs().max().item())
        assert (fg.float().cpu() - fw).abs().max() < 5e-3 * max(1.0, fw.abs().max().item())
        s, rgb2, sem = model(x.cuda(), d.cuda())
        assert s.shape == (500,) and rgb2.shape == (500, 3) and torch.allclose(sem.sum(-1), torch.ones(500).cuda(), atol=1e-3)


def test_unknown_encoding_and_cpu_model_raise():
    from autolabel_amd.models import ALNetwork
    with pytest.raises(NotImplementedError):
        ALNetwork(encoding='nope')
    m = ALNetwork(encoding='hg+freq', hidden_dim=128, hidden_dim_color=128, num_layers_color=2)
    with pytest.raises(RuntimeError):
        m.density(torch.zeros(4, 3))  # CPU model: no fallback


def test_render_staged_shapes_and_state_dict_roundtrip():
    model = make_model().eval()
    H_, W_ = 12, 20
    g = torch.Generator().manual_seed(1)
    o = torch.zeros(H_, W_, 3).cuda()
    d = torch.nn.functional.normalize(torch.randn(H_, W_, 3, generator=g), dim=-1).cuda()
    n = torch.ones(H_ * W_, 1).cuda()  # the [H*W,1] quirk of _get_test
    with torch.inference_mode():
        out = model.render(o, d, n, staged=True, perturb=False, num_steps=64, upsample_steps=0, max_ray_batch=100,
                           rgb_weight=1.0, feature_loss=False)  # extra opt entries are ignored like **vars(opt)
    assert out['image'].shape == (H_, W_, 3) and out['depth'].shape == (H_, W_)
    assert out['semantic'].shape == (H_, W_, 3) and out['semantic_features'].shape == (H_, W_, 64)
    assert out['depth_variance'].shape == (H_, W_) and out['coordinates_map'].shape == (H_, W_, 3)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert 'sigma_net.params' in sd and 'encoder.grid_encoding.params' in sd
    m2 = make_model(grid_scale=1.0)
    m2.load_state_dict(sd)
    with torch.inference_mode():
        out2 = m2.eval().render(o, d, n, staged=True, perturb=False, num_steps=64, upsample_steps=0)
    assert torch.equal(out['image'], out2['image'])


def _opt(feature_loss=True):
    return Namespace(rand_pose=-1, color_space='srgb', feature_loss=feature_loss, rgb_weight=1.0, depth_weight=0.1,
                     semantic_weight=1.0, feature_weight=0.5)


def _trainer(model, fused, fp16=True):
    from autolabel_amd.trainer import SimpleTrainer
    optimizer = lambda model: torch.optim.Adam([{'name': 'encoding', 'params': list(model.encoder.parameters())},
                                                {'name': 'net', 'params': model.network_parameters(), 'weight_decay': 1e-6}],
                                               lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    sched = lambda o: torch.optim.lr_scheduler.StepLR(o, gamma=0.5, step_size=1)
    return SimpleTrainer('ngp', _opt(), model, device='cuda:0', workspace=None, optimizer=optimizer,
                         criterion=torch.nn.MSELoss(reduction='none'), fp16=fp16, ema_decay=0.95, lr_scheduler=sched,
                         scheduler_update_every_step=False, metrics=[], use_checkpoint='latest', fused=fused, mute=True)


def _host_batches(n, B=512):
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import ArrayDataset
    import random
    scene = synthetic.make_room_scene(n_frames=6, w=64, h=48, fx=32.0, fy=32.0, cx=31.5, cy=23.5, feat_dim=16, feat_hw=(6, 8),
                                      labelled_every=2)
    ds = ArrayDataset(scene, batch_size=B)
    np.random.seed(0); random.seed(0)
    return ds, [ds._next_train() for _ in range(n)]


def test_generic_and_fused_training_steps_agree():
    """Reference-shaped loop (render -> torch loss -> GradScaler -> torch Adam) vs the fused engine, same batches."""
    ds, batches = _host_batches(3)
    bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
    finals = []
    for fused in (False, True):
        torch.manual_seed(0)
        model = make_model(D=64, C_=7, bound=bound, grid_scale=1.0)
        tr = _trainer(model, fused)
        assert tr.fused == fused
        tr.train_iterations(iter(batches), 3)
        finals.append(torch.cat([p.detach().reshape(-1).cpu() for _, p in model._param_blocks()]))
        assert tr.optimizer.param_groups[0]['lr'] == 2.5e-3  # StepLR stepped once after the 3 iterations
    a, b = finals
    n_grid = model._layout.n_grid
    moved = (b[:n_grid] - make_model(D=64, C_=7, bound=bound, grid_scale=1.0).encoder.grid_encoding.params.detach().cpu()).abs() > 0
    assert moved.sum() > 1000
    # Adam normalises the step size: compare the updates. fp16 gradient paths are identical kernels in both modes;
    # differences come from torch's GradScaler/Adam op order only.
    assert (a[n_grid:] - b[n_grid:]).abs().max() < 2e-3
    # grid: Adam (eps 1e-15) turns ANY non-zero gradient into a +-lr step, so entries whose gradient is ~0 flip sign with
    # the (non-deterministic) order of the fp32 atomics -- even between two runs of the same mode; bound their share
    assert ((a[:n_grid] - b[:n_grid]).abs() > 2e-3).float().mean() < 2e-2


def test_engine_follows_checkpoint_load_and_device_rebind(tmp_path):
    """the fused engine must train the buffers the model renders from.  (1) a checkpoint loaded AFTER the engine
    exists restores the weights, the fp16 shadows and the Adam state; (2) model.cuda() re-binds the parameters to a new flat
    buffer -- the next step must move THAT buffer and keep the optimizer state."""
    from autolabel_amd.trainer import SimpleTrainer
    ds, batches = _host_batches(6)
    bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
    torch.manual_seed(0)
    model = make_model(D=64, C_=7, bound=bound, grid_scale=1.0)
    tr = _trainer(model, True)
    tr.workspace, tr.ckpt_path = str(tmp_path), str(tmp_path / 'checkpoints')
    os.makedirs(tr.ckpt_path, exist_ok=True)
    tr.train_iterations(iter(batches[:3]), 3)
    tr.save_checkpoint('a')
    trained = torch.cat([p.detach().reshape(-1).clone() for _, p in model._param_blocks()])
    m_state = tr.engine.m.clone()
    t0 = ds._get_test(0)
    rays = lambda: (torch.as_tensor(t0['rays_o']).cuda(), torch.as_tensor(t0['rays_d']).cuda(), torch.as_tensor(t0['direction_norms']).cuda())
    with torch.inference_mode():
        ref = model.eval().render(*rays(), staged=True, perturb=False, num_steps=32, upsample_steps=0)['image'].clone()
    # keep training, then go back to the checkpoint with the engine alive
    tr.train_iterations(iter(batches[3:]), 3)
    assert not torch.equal(torch.cat([p.detach().reshape(-1) for _, p in model._param_blocks()]), trained)
    tr.load_checkpoint(str(tmp_path / 'checkpoints' / 'a.pth'))
    eng = tr._engine()
    assert torch.equal(eng.m, m_state) and tr._engine_state is None
    with torch.inference_mode():
        again = model.eval().render(*rays(), staged=True, perturb=False, num_steps=32, upsample_steps=0)['image']
    assert torch.equal(again, ref), 'render after load_checkpoint must use the restored weights (stale fp16 shadows?)'
    # re-bind: nn.Module._apply drops the flat buffer; the engine must follow and carry m / v / step counters
    steps_before = int(eng.state_i[0].item())
    model.cuda()
    model.float()
    tr.train_iterations(iter(batches[3:4]), 1)
    assert tr.engine is not eng and tr.engine.pipe is model._pipe
    assert int(tr.engine.state_i[0].item()) == steps_before + 1
    now = torch.cat([p.detach().reshape(-1) for _, p in model._param_blocks()])
    assert not torch.equal(now, trained), 'the step after the re-bind must train the buffer the model renders from'
    assert now.data_ptr() != 0 and model.sigma_net.params.data_ptr() == model._P.flat[model._layout.offsets['sigma']:].data_ptr()


def _graph_vs_eager(render_between):
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames, DeviceLoader
    scene = synthetic.make_cube_scene(n_frames=8)
    finals = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        model = make_model(D=64, C_=scene['n_classes'], bound=6.0, grid_scale=1.0)
        tr = _trainer(model, True)
        tr.use_graph = use_graph
        tr.opt.feature_loss = False
        frames = DeviceFrames.from_scene(scene, 'cuda')
        loader = DeviceLoader(frames, 1024, 1000, seed=3)
        tr.train_iterations(loader, 3)
        if render_between:
            # a render of another shape through the same model: with a name-keyed workspace shared between training and rendering
            # this reallocated the buffers the captured step points into (use-after-free on replay)
            model.eval()
            with torch.no_grad():
                o = torch.zeros(777, 3, device='cuda'); d = torch.nn.functional.normalize(torch.randn(777, 3, device='cuda'), dim=1)
                out = model.render(o, d, torch.ones(777, 1, device='cuda'), staged=True, max_ray_batch=500, num_steps=96, upsample_steps=32)
            assert torch.isfinite(out['image']).all()
        tr.train_iterations(loader, 3)     # lr halves in between (StepLR): a device word, the capture is reused
        assert tr.global_step == 6 and loader.step == 6
        assert int(tr.engine.state_i[0].item()) == 6
        finals.append((torch.cat([p.detach().reshape(-1).cpu() for _, p in model._param_blocks()]), tr.engine.terms.cpu().clone()))
    return finals


def test_graphed_training_loop_matches_launch_by_launch():
    """SimpleTrainer.train_iterations over a dataset.DeviceLoader replays ONE captured hipGraph per step; same seeds and step
    numbers as the launch-by-launch loop (use_graph=False).  Every kernel of the step is order-independent (integer-accumulated
    scatter, slab-reduced weight gradients, row-order compaction), so the two trajectories are BIT-IDENTICAL."""
    (a, ta), (b, tb) = _graph_vs_eager(render_between=False)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), f'{(a != b).sum().item()} of {a.numel()} parameters differ between graph replay and launch-by-launch'
    assert abs(ta[4].item() - tb[4].item()) < 1e-5 * max(tb[4].item(), 1e-3)   # (the reported loss terms meet in fp32 atomics)


def test_render_between_graph_replays_cannot_move_the_captured_buffers():
    """train -> render with other shapes -> train on the graph path equals the launch-by-launch loop bit for bit: the engine's
    step lives in its own workspace (TrainEngine.ws), renders in the pipeline's."""
    (a, _), (b, _) = _graph_vs_eager(render_between=True)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_training_is_bit_reproducible_run_to_run():
    """Two fresh processes' worth of state in one process: same seeds, 12 optimizer steps each through the fused engine at a batch
    that spans many tiles; every parameter must agree bit for bit (dense path and occupancy-grid marching)."""
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    scene = synthetic.make_cube_scene(n_frames=8)
    frames = DeviceFrames.from_scene(scene, 'cuda')
    for march in (False, True):
        finals = []
        for _ in range(2):
            layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=6.0)
            P = Params(layout, torch.device('cuda'))
            P.init_(seed=0)
            pipe = HipPipeline(layout, P)
            if march:
                pipe.enable_marching(G=64, max_steps=256, samples=48)
                pipe.mark_untrained_grid(frames.world_to_camera(), (frames.desc.fx, frames.desc.fy, frames.desc.cx, frames.desc.cy),
                                         size=(frames.w, frames.h))
            eng = TrainEngine(pipe, num_steps=48, upsample_steps=48)
            batch = frames.alloc_batch(2048)
            for it in range(12):
                frames.next_train(batch, seed=5, step=it)
                eng.step(batch, seed=9, step=it)
            torch.cuda.synchronize()
            assert int(eng.state_i[0].item()) > 0
            finals.append((P.flat.clone(), eng.m.clone(), eng.v.clone()))
        for x, y in zip(*finals):
            assert torch.equal(x, y), f'march={march}: {(x != y).sum().item()} of {x.numel()} values differ between two identical runs'


def test_lseg_training_is_bit_reproducible_run_to_run():
    """The LSeg-width heads (512-d features, wide.hip): their weight-gradient GEMM adds per-row-range partial sums in a fixed order
    (rounds 1-2 flushed them with fp32 atomics), so two identical runs agree bit for bit -- with the semantic loss (every GEMM per
    sample) and with semantic_weight = 0 (the linear last layer per ray)."""
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    scene = synthetic.make_room_scene(n_frames=8, w=64, h=48, fx=32.0, fy=32.0, cx=31.5, cy=23.5, feat_dim=64, feat_hw=(6, 8), labelled_every=2)
    frames = DeviceFrames.from_scene(scene, 'cuda')
    for sem_w in (1.0, 0.0):
        finals = []
        for _ in range(2):
            layout = ModelLayout('hg+freq', 15, 128, 128, 512, scene['n_classes'], bound=6.0)
            P = Params(layout, torch.device('cuda'))
            P.init_(seed=0)
            eng = TrainEngine(HipPipeline(layout, P), num_steps=32, upsample_steps=32, feature_loss=True, semantic_weight=sem_w)
            assert eng.sem_linear == (sem_w == 0.0)
            batch = frames.alloc_batch(1024)
            for it in range(6):
                frames.next_train(batch, seed=5, step=it)
                eng.step(batch, seed=9, step=it)
            torch.cuda.synchronize()
            assert int(eng.state_i[0].item()) > 0
            finals.append((P.flat.clone(), eng.m.clone(), eng.v.clone()))
        for x, y in zip(*finals):
            assert torch.equal(x, y), f'semantic_weight={sem_w}: {(x != y).sum().item()} of {x.numel()} values differ between two identical runs'


def _fused_vs_separate(frames, scene, march, batch, samples, steps, scale):
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    engs = []
    for fuse in (True, False):
        layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=6.0)
        P = Params(layout, torch.device('cuda'))
        P.init_(seed=0)
        eng = TrainEngine(HipPipeline(layout, P), num_steps=samples, upsample_steps=samples, fuse_grid_adam=fuse, scaler={'init_scale': scale})
        engs.append((eng, P, frames.alloc_batch(batch)))
    for it in range(steps):
        for eng, P, b in engs:
            frames.next_train(b, seed=5, step=it)
            eng.step(b, seed=9, step=it)
        (ea, Pa, _), (eb, Pb, _) = engs
        for name, x, y in [('params', Pa.flat, Pb.flat), ('m', ea.m, eb.m), ('v', ea.v, eb.v), ('table16', Pa.table16, Pb.table16),
                           ('state_i', ea.state_i, eb.state_i)]:
            assert torch.equal(x, y), f'batch {batch} step {it}: {name} differs in {(x != y).sum().item()} of {x.numel()} values'
    assert int(engs[0][0].state_i[0].item()) == steps
    del engs
    torch.cuda.empty_cache()


def test_grid_adam_inside_the_scatter_equals_the_separate_optimizer_bit_for_bit():
    """On one GPU the scatter's second phase takes the Adam step for the hash table itself (TrainEngine.fuse_grid_adam); the
    gradient-through-HBM route (what a data-parallel run uses: P.grad, then aln_adam_step over everything) must leave every
    parameter, moment, fp16 shadow and optimizer / GradScaler state word identical after every step -- including steps the loss
    scaler skips (started at 2^36 the first steps overflow) and a change of learning rate; dense path and marching."""
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    scene = synthetic.make_cube_scene(n_frames=8)
    frames = DeviceFrames.from_scene(scene, 'cuda')
    _fused_vs_separate(frames, scene, march=False, batch=4096, samples=128, steps=3, scale=2.0 ** 10)   # the bench's launch sizes (2^20 rows)
    for march in (False, True):
        engs = []
        for fuse in (True, False):
            layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=6.0)
            P = Params(layout, torch.device('cuda'))
            P.init_(seed=0)
            pipe = HipPipeline(layout, P)
            if march:
                pipe.enable_marching(G=64, max_steps=256, samples=48)
                pipe.mark_untrained_grid(frames.world_to_camera(), (frames.desc.fx, frames.desc.fy, frames.desc.cx, frames.desc.cy),
                                         size=(frames.w, frames.h))
            eng = TrainEngine(pipe, num_steps=48, upsample_steps=48, fuse_grid_adam=fuse, scaler={'init_scale': 2.0 ** 36, 'growth_interval': 5})
            assert eng.fuse_grid_adam == fuse
            engs.append((eng, P, frames.alloc_batch(1024)))
        for it in range(24):
            for eng, P, batch in engs:
                if it == 19:
                    eng.lr = 2e-3
                frames.next_train(batch, seed=5, step=it)
                eng.step(batch, seed=9, step=it)
            (ea, Pa, _), (eb, Pb, _) = engs
            for name, x, y in [('params', Pa.flat, Pb.flat), ('m', ea.m, eb.m), ('v', ea.v, eb.v), ('table16', Pa.table16, Pb.table16),
                               ('state_i', ea.state_i, eb.state_i), ('state_f', ea.state_f, eb.state_f)]:
                assert torch.equal(x, y), f'march={march} step {it}: {name} differs in {(x != y).sum().item()} of {x.numel()} values'
        assert 4 < int(ea.state_i[0].item()) < 24, f'the run must contain applied AND skipped steps ({int(ea.state_i[0].item())} applied)'
        assert Pa.grad[:ea.L.n_grid].abs().max().item() == 0, 'the fused route never writes the table gradient'


def test_training_reduces_loss_on_cube_scene():
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    scene = synthetic.make_cube_scene()
    frames = DeviceFrames.from_scene(scene, 'cuda')
    model = make_model(D=64, C_=3, bound=6.0, grid_scale=1.0)
    eng = TrainEngine(model._ensure_device(), num_steps=64, upsample_steps=64)
    batch = frames.alloc_batch(2048)
    losses = []
    for i in range(300):
        frames.next_train(batch, seed=1, step=i)
        eng.step(batch, seed=2, step=i)
        if i % 50 == 0 or i == 299:
            losses.append(eng.terms.tolist())
    assert all(math.isfinite(v) for t in losses for v in t)
    # (12 runs, scripts/dev/debug_dense_psnr.py: ratio 0.206 - 0.231, PSNR 16.5 - 17.7 dB)
    assert losses[-1][0] < 0.3 * losses[0][0], f'rgb loss did not drop: {losses[0]} -> {losses[-1]}'
    assert losses[-1][1] < 0.5 * losses[0][1], 'depth loss did not drop'
    # rendering a training view reproduces it
    t = frames.get_test(0)
    with torch.inference_mode():
        out = model.eval().render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False, num_steps=128, upsample_steps=0)
    psnr = -10 * math.log10(((out['image'] - t['pixels']) ** 2).mean().item())
    assert psnr > 15.0, psnr


def test_scripts_train_export_on_written_scene(tmp_path):
    """scripts/train.py -> checkpoint + params.pkl -> scripts/export.py on a scene directory in the reference's layout."""
    from autolabel_amd import synthetic
    from autolabel_amd.utils import write_scene
    scene_dir = str(tmp_path / 'scene1')
    write_scene(synthetic.make_cube_scene(n_frames=6), scene_dir)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train.py'), scene_dir, '--iters', '1000', '--batch-size', '1024',
                        '--factor-train', '1', '--workers', '0', '--eval', '--factor-test', '1'], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    mdir = os.path.join(scene_dir, 'nerf', 'g15_hg+freq_plain_rgb1.0_d0.1_s1.0_f0.5')
    assert os.path.exists(os.path.join(mdir, 'params.pkl')) and os.listdir(os.path.join(mdir, 'checkpoints'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'export.py'), scene_dir], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = sorted(os.listdir(os.path.join(scene_dir, 'output', 'semantic')))
    assert len(outs) == 6


def test_scripts_train_with_marching_and_device_data(tmp_path):
    """scripts/train.py --cuda-ray --device-data: frames in HBM, occupancy-grid marching, the step replayed from the two
    hipGraphs (plain / with grid refresh); export.py rebuilds the marching model from params.pkl and loads grid + bitfield."""
    from autolabel_amd import synthetic
    from autolabel_amd.utils import write_scene
    scene_dir = str(tmp_path / 'scene2')
    write_scene(synthetic.make_cube_scene(n_frames=6), scene_dir)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train.py'), scene_dir, '--iters', '1000', '--batch-size', '1024',
                        '--factor-train', '1', '--workers', '0', '--cuda-ray', '--march-samples', '64', '--device-data'],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    mdir = os.path.join(scene_dir, 'nerf', 'g15_hg+freq_plain_rgb1.0_d0.1_s1.0_f0.5')
    ck = sorted(os.listdir(os.path.join(mdir, 'checkpoints')))
    sd = torch.load(os.path.join(mdir, 'checkpoints', ck[-1]), map_location='cpu', weights_only=False)['model']
    assert 'density_grid' in sd and (sd['density_grid'] > 0).any() and (sd['density_bitfield'] != 0).any()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'export.py'), scene_dir], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(os.listdir(os.path.join(scene_dir, 'output', 'semantic'))) == 6


@pytest.mark.parametrize('encoding', ['freq', 'hg'])
def test_other_encodings_train_one_step_like_the_oracle(encoding):
    """--encoding freq / hg (autolabel/model_utils.py:25-29): forward + gradients vs the oracle, then one fused step."""
    from autolabel_amd.engine import TrainEngine
    from test_gpu_pipeline import _batch, flat_grads, hip_loss, make_rays, rel
    model = make_model(encoding=encoding, bound=1.0, C_=3, grid_scale=3e3)
    oracle, cfg = oracle_of(model)
    pipe = model._ensure_device()
    N, S1, S2 = 48, 32, 32
    o, d, norms = make_rays(N, seed=4)
    g = torch.Generator().manual_seed(9)
    noise, u = torch.rand(N, S1, generator=g), torch.rand(N, S2, generator=g)
    batch = _batch(N, 3, 16, seed=2)
    od, dd, nd, nz, ud = o.cuda(), d.cuda(), norms.cuda().reshape(-1), noise.cuda(), u.cuda()
    out, ctx = pipe.forward(od, dd, nd, S1, S2, True, train=True, noise=nz, u=ud)
    want = oracle.run(o, d, norms, num_steps=S1, upsample_steps=S2, perturb=True, noise_coarse=noise, u_fine=u,
                      z_fine_override=ctx['z'][N * S1:].view(N, S2).cpu())
    assert (out['image'].cpu() - want['image']).abs().max() < 5e-3
    loss, _ = O.loss_fn(want, batch, feature_loss=True)
    loss.backward()
    gi, gd, gs, gf, t = hip_loss(pipe, out, batch, N, 3, 64, 16, scale=512.0)
    pipe.P.grad.zero_()
    pipe.backward(ctx, gi, gd, gs, gf)
    got = pipe.P.grad[:pipe.L.n_total].cpu() / 512.0
    assert rel(got, flat_grads(oracle, cfg)) < 1.5e-2
    eng = TrainEngine(pipe, num_steps=S1, upsample_steps=S2, feature_loss=True)
    before = pipe.P.flat.clone()
    dev = {k: v.cuda().float().contiguous() for k, v in batch.items() if k != 'semantic'}
    dev.update(rays_o=od, rays_d=dd, direction_norms=nd, semantic=batch['semantic'].int().cuda())
    eng.step(dev, seed=1, step=0)
    assert eng.state_i[0].item() == 1 and not torch.equal(before, pipe.P.flat) and torch.isfinite(pipe.P.flat).all()


def test_host_loader_over_a_dataset_is_replaced_by_device_resident_frames():
    """The reference's route (scripts/train.py:65-93: DataLoader over the dataset -> SimpleTrainer.train): the trainer moves the
    frames into HBM by itself and replays the step from the hipGraph; device_data=False keeps the host loader; label edits made
    on the host dataset reach the device copy through refresh_resident_labels (InteractiveTrainer.dataset_updated)."""
    from autolabel_amd.dataset import DeviceLoader, LenDataset
    ds, _ = _host_batches(1, B=512)
    bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
    losses = {}
    for mode in ('auto', False):
        torch.manual_seed(0)
        model = make_model(D=64, C_=7, bound=bound, grid_scale=1.0)
        tr = _trainer(model, True)
        tr.device_data = mode
        loader = torch.utils.data.DataLoader(LenDataset(ds, 1000), batch_size=None, num_workers=0)
        loader._data = ds
        tr.train_iterations(loader, 6)
        used = tr.resident_loader(loader)
        assert isinstance(used, DeviceLoader) == (mode == 'auto')
        assert tr.global_step == 6 and int(tr.engine.state_i[0]) == 6
        losses[mode] = float(tr.engine.terms[4])
        assert np.isfinite(losses[mode])
        if mode == 'auto':
            assert used.frames.n_frames == ds.n_examples and used.batch['rays_o'].shape == (512, 3) and used._data is ds
            ds.semantics[1, :100] = 3     # a label edit on the host (SceneDataset.semantic_map_updated)
            ds.index_sampler.update(ds.semantics)
            tr.refresh_resident_labels()         # what InteractiveTrainer.dataset_updated calls
            assert int(used.frames.semantics.reshape(ds.n_examples, -1)[1, :100].min()) == 3
            ds.semantics[1, :100] = 0
            ds.index_sampler.update(ds.semantics)


def test_label_edits_on_the_host_dataset_reach_the_captured_step():
    """The reference's label-edit flows only touch the host dataset (scripts/simulate_user.py:89 update_sampler(), backend.py:155
    semantic_map_updated()) and never tell the trainer.  The device-resident copy follows the IndexSampler's update count, and the
    captured step -- which holds the class-index pointers and the class count BY VALUE -- is captured again."""
    from autolabel_amd.dataset import DeviceLoader, LenDataset
    ds, _ = _host_batches(1, B=512)
    ds.semantics[:] = 0                       # no labels at all to begin with
    ds.index_sampler.update(ds.semantics)
    bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
    torch.manual_seed(0)
    tr = _trainer(make_model(D=64, C_=7, bound=bound, grid_scale=1.0), True)
    loader = torch.utils.data.DataLoader(LenDataset(ds, 1000), batch_size=None, num_workers=0)
    loader._data = ds
    tr.train_iterations(loader, 3)
    used = tr.resident_loader(loader)
    assert isinstance(used, DeviceLoader) and tr._graph is not None
    g0 = tr._graph[2]
    assert int(used.frames.desc.n_classes) == 0 and bool((used.batch['semantic'] == -1).all())
    ds.semantics[1, :100] = 3                 # the user paints 100 pixels of frame 1 (class 3 -> label 2 in the batch) ...
    ds.index_sampler.update(ds.semantics)     # ... and the host flow refreshes the sampler; nobody calls the trainer
    seen = False
    for _ in range(8):                        # a 512-ray batch is one chunk: class-weighted with probability 0.5 (dataset.py:207-211)
        tr.train_iterations(loader, 1)
        seen |= bool((used.batch['semantic'] == 2).all())
    assert tr.resident_loader(loader) is used and tr._graph[2] is not g0, 'the step must be captured again after the class index moved'
    assert int(used.frames.desc.n_classes) == 1 and int(used.frames.semantics.reshape(ds.n_examples, -1)[1, :100].min()) == 3
    assert seen, 'no class-weighted chunk drew from the newly labelled pixels in 8 steps'
    assert np.isfinite(float(tr.engine.terms[4])) and int(tr.engine.state_i[0]) == 11


def test_interactive_trainer_and_eval_steps():
    """InteractiveTrainer.init/take_step (GUI / ROS loop) and SimpleTrainer.test_step / eval_step / evaluate shapes."""
    from autolabel_amd.trainer import InteractiveTrainer
    ds, batches = _host_batches(4, B=512)
    bound = float(((ds.max_bounds - ds.min_bounds) - (ds.min_bounds + ds.max_bounds) * 0.5).max())
    model = make_model(D=64, C_=7, bound=bound, grid_scale=1.0)
    optimizer = lambda model: torch.optim.Adam([{'name': 'encoding', 'params': list(model.encoder.parameters())},
                                                {'name': 'net', 'params': model.network_parameters(), 'weight_decay': 1e-6}],
                                               lr=5e-3, betas=(0.9, 0.99), eps=1e-15)
    tr = InteractiveTrainer('ngp', _opt(feature_loss=False), model, device='cuda:0', workspace=None, optimizer=optimizer,
                            criterion=torch.nn.MSELoss(reduction='none'), fp16=True, ema_decay=0.95,
                            lr_scheduler=lambda o: torch.optim.lr_scheduler.StepLR(o, gamma=0.5, step_size=1), metrics=[],
                            use_checkpoint='latest', mute=True)

    class Loader:
        _data = ds

        def __iter__(self):
            return iter(batches)
    tr.init(Loader())
    l0 = float(tr.take_step())
    for _ in range(3):
        l1 = float(tr.take_step())
    assert tr.step == 4 and np.isfinite(l0) and np.isfinite(l1)
    ds.split = 'test'
    frame = ds._get_test(0)
    with torch.no_grad():
        rgb, depth, sem, feat = tr.test_step(frame)
        assert rgb.shape == (1, 48, 64, 3) and depth.shape == (1, 48, 64) and sem.shape == (1, 48, 64, 7) and feat.shape == (48, 64, 64)
        p_rgb, p_depth, p_sem, gt_rgb, loss = tr.eval_step(frame)
        assert p_rgb.shape == (1, 48, 64, 3) and p_sem.shape == (1, 48, 64, 7) and torch.isfinite(loss)
    assert np.isfinite(tr.evaluate([ds._get_test(i) for i in range(2)]))


def test_render_options_bg_color_and_chunking_consistency():
    model = make_model().eval()
    g = torch.Generator().manual_seed(2)
    o = ((torch.rand(300, 3, generator=g) - 0.5) * 0.4).cuda()
    d = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1).cuda()
    n = torch.ones(300, 1).cuda()
    with torch.inference_mode():
        a = model.render(o, d, n, staged=False, perturb=False, num_steps=64, upsample_steps=32)
        b = model.render(o, d, n, staged=True, perturb=False, num_steps=64, upsample_steps=32, max_ray_batch=77)
        c = model.render(o, d, n, staged=False, perturb=False, num_steps=64, upsample_steps=32, bg_color=0.0)
    for k in ['image', 'depth', 'semantic', 'semantic_features', 'weights_sum']:
        assert torch.allclose(a[k], b[k], atol=1e-6), k          # chunking does not change per-ray results
    assert torch.allclose(a['image'] - c['image'], (1 - a['weights_sum'])[:, None].expand(-1, 3), atol=1e-6)


def test_open_vocabulary_queries_follow_the_reference_recipe():
    """autolabel/evaluation.py:295-327 (pixels) and :400-445 (points): feature . text argmax, restated with plain loops."""
    from autolabel_amd import queries as Q
    model = make_model(D=64, C_=3, bound=1.5)
    oracle, cfg = oracle_of(model)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(700, 3, generator=g) * 2 - 1) * 1.4
    text = torch.nn.functional.normalize(torch.randn(5, 64, generator=g), dim=1)
    # point features == semantic(density(x).geo_feat)[1] of the oracle
    with torch.no_grad():
        want_f = oracle.semantic(oracle.density(pts)['geo_feat'])[1]
    got_f = Q.point_features(model, pts.cuda()).cpu()
    assert got_f.shape == (700, 64)
    assert (got_f - want_f).abs().max() < 2e-2 * max(1.0, want_f.abs().max().item())
    # one evaluation (no jitter): the reference loop over prompts
    got = Q.predict_semantic_points(model, pts.cuda(), text.cuda(), n_evals=1, batch_size=256).cpu()
    fn = got_f / torch.norm(got_f, dim=-1, keepdim=True)
    sims = torch.stack([(fn * text[i][None]).sum(dim=-1) for i in range(5)], dim=1)
    top2 = sims.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3                       # ties within fp noise may flip
    assert torch.equal(got[clear], sims.argmax(dim=1)[clear]) and clear.float().mean() > 0.9
    # jittered averaging: weights 1 + (n - 1) x 1/n, label map applied
    gen = torch.Generator(device='cuda').manual_seed(11)
    lab = torch.tensor([10, 11, 12, 13, 14], device='cuda')
    out = Q.predict_semantic_points(model, pts.cuda(), text.cuda(), label_id_map=lab, n_evals=4, generator=gen)
    assert out.shape == (700,) and int(out.min()) >= 10 and int(out.max()) <= 14
    gen2 = torch.Generator(device='cuda').manual_seed(11)
    f = Q.point_features(model, pts.cuda())
    for _ in range(3):
        f += Q.point_features(model, pts.cuda() + torch.randn(pts.shape, device='cuda', generator=gen2) * 0.02) * 0.25
    assert torch.equal(out, lab[Q.similarity_argmax(f, text.cuda())])
    # zero feature rows argmax to class 0 like the NaN rows of the reference
    z = torch.zeros(3, 64, device='cuda'); z[1] = text[3].cuda()
    assert Q.similarity_argmax(z, text.cuda()).tolist() == [0, 3, 0]
    # pixels: staged render + argmax
    H_, W_ = 6, 8
    o = torch.zeros(H_, W_, 3); o[..., 2] = -1.2
    d = torch.nn.functional.normalize(torch.stack(torch.meshgrid(torch.linspace(-.4, .4, H_), torch.linspace(-.4, .4, W_), indexing='ij') +
                                                  (torch.ones(H_, W_),), dim=-1), dim=-1)
    batch = {'rays_o': o.numpy(), 'rays_d': d.numpy(), 'direction_norms': np.ones((H_ * W_, 1), np.float32)}
    img = Q.predict_semantic_image(model, batch, text.cuda(), num_steps=64, upsample_steps=32)
    assert img.shape == (H_, W_)
    feats = model.render(o.cuda(), d.cuda(), torch.ones(H_ * W_, 1, device='cuda'), staged=True, perturb=False, num_steps=64,
                         upsample_steps=32)['semantic_features']
    assert torch.equal(img, Q.similarity_argmax(feats, text.cuda()))


def test_reference_checkpoint_import_and_fused_grid_position(tmp_path):
    """a `checkpoints/*.pth` in the reference's format, WRITTEN BY THE ORACLE from its own per-layer matrices
    (oracle/tcnn_pack.py: fp16 grid, padded [out, in] matrices, 1-D and 2-D tensors, fork-only buffers) -- not derived from the
    model under test -- loads through model_utils.load_checkpoint(reference=True); density / color / semantic of the loaded
    model then match the oracle evaluated on the same matrices, the grid position follows tcnn's fused multiply-add and the
    hash-grid features are bit-exact to the oracle evaluated the same way."""
    import ctypes as C
    from autolabel_amd import hip as H, model_utils
    from oracle.tcnn_pack import pack_reference_state_dict
    bound = 2.0
    cfg = O.ModelConfig(feature_dim=64, n_classes=5, bound=bound, grid=O.GridSpec(pos_fma=True))
    params = O.init_params(cfg, seed=11)
    with torch.no_grad():
        params['grid'].mul_(2e3)                          # a trained-looking table (the initialisation is +-1e-4)
        params['grid'].copy_(params['grid'].half().float())   # what an fp16 grid copy holds
    ref_sd = pack_reference_state_dict(params, cfg, bound=bound)
    assert ref_sd['color_net.params'].dim() == 2 and ref_sd['sigma_net.params'].dim() == 1
    assert ref_sd['encoder.grid_encoding.params'].dtype == torch.float16
    os.makedirs(tmp_path / 'checkpoints')
    torch.save({'model': ref_sd, 'epoch': 10}, tmp_path / 'checkpoints' / 'ngp_ep0010.pth')
    m2 = make_model(D=64, C_=5, bound=bound, grid_scale=1.0)
    assert not m2.tcnn_fma
    model_utils.load_checkpoint(m2, str(tmp_path / 'checkpoints'), reference=True)
    assert m2.tcnn_fma and m2._layout.enc.grid.pos_fma == 1
    assert torch.equal(m2.encoder.grid_encoding.params.detach().cpu(), params['grid'].reshape(-1))
    # a wrong head width is refused with both counts in the message
    bad = dict(ref_sd); bad['semantic_out.params'] = torch.zeros(7)
    with pytest.raises(ValueError, match='semantic_out.params'):
        model_utils.import_reference_state_dict(make_model(D=64, C_=5, bound=bound), bad)
    # point queries of the loaded model against the oracle on ITS matrices (models.py:175-220, 248-256)
    oracle = O.OracleModel(cfg, params={k: v.clone() for k, v in params.items()}, half_sim=True)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(4000, 3, generator=g) * 2 - 1) * bound
    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=1)
    with torch.no_grad():
        want = oracle.density(x)
        want_rgb = oracle.color(x, d, geo_feat=want['geo_feat'])
        want_logits, want_f = oracle.semantic(want['geo_feat'])
        m2.eval()
        got = m2.density(x.cuda())
        got_rgb = m2.color(x.cuda(), d.cuda(), geo_feat=got['geo_feat'])
        got_logits, got_f = m2.semantic(got['geo_feat'])
    tol = lambda t: 4e-3 * max(1.0, t.abs().max().item())
    assert (got['geo_feat'].float().cpu() - want['geo_feat']).abs().max() <= tol(want['geo_feat'])
    assert (got['sigma'].cpu().log() - want['sigma'].log()).abs().max() <= 4e-3 * max(1.0, want['sigma'].log().abs().max().item())
    assert (got_rgb.cpu() - want_rgb).abs().max() <= 4e-3
    assert (got_f.float().cpu() - want_f).abs().max() <= tol(want_f) and (got_logits.float().cpu() - want_logits).abs().max() <= tol(want_logits)
    # features: HIP (fma on) == oracle (pos_fma=True) bit for bit; with the switch off a few last-bit cases differ
    want_enc = oracle.encode(x)[:, 12:].half()
    pipe, e = m2._ensure_device(), m2._layout.enc
    enc = torch.zeros(4000, e.enc_pad, dtype=torch.float16, device='cuda')
    xd = x.cuda().contiguous()
    H.call('aln_encode_fwd', C.byref(e), H.ptr(pipe.P.table16), None, None, None, H.ptr(xd), 4000, 1, H.ptr(enc), H.stream())
    assert torch.equal(enc.cpu()[:, 12:44], want_enc)
    m2.set_tcnn_fma(False)
    H.call('aln_encode_fwd', C.byref(e), H.ptr(pipe.P.table16), None, None, None, H.ptr(xd), 4000, 1, H.ptr(enc), H.stream())
    assert not torch.equal(enc.cpu()[:, 12:44], want_enc)


def test_backward_through_an_overwritten_render_context_raises():
    """intermediates live in shared workspaces; a second render() before backward() used to give silently wrong
    gradients.  Now the stale context is detected."""
    model = make_model().train()
    g = torch.Generator().manual_seed(0)
    o = torch.zeros(64, 3).cuda()
    d = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1).cuda()
    n = torch.ones(64, 1).cuda()
    a = model.render(o, d, n, perturb=True, num_steps=32, upsample_steps=32)
    with torch.no_grad():
        model.render(o, d, n, perturb=False, num_steps=32, upsample_steps=0)      # e.g. a preview in between
    with pytest.raises(RuntimeError, match='another forward'):
        a['image'].sum().backward()
    b = model.render(o, d, n, perturb=True, num_steps=32, upsample_steps=32)      # the usual order still works
    b['image'].sum().backward()
    assert model.sigma_net.params.grad is not None and torch.isfinite(model.sigma_net.params.grad).all()
    with pytest.raises(NotImplementedError, match='num_layers_color=2'):
        from autolabel_amd.models import ALNetwork
        ALNetwork(encoding='hg+freq', hidden_dim=128, hidden_dim_color=128)    # class default num_layers_color=3


def test_graph_replays_survive_kernel_launches_in_between():
    """ROCm 7.2: with hipGraph packet capture enabled (the default), an ordinary kernel launch between two replays of a captured
    step leaves the next replay with stale kernel arguments (NaN losses or a memory access fault) -- reproduced with
    scripts/dev/debug_graph_eager.py.  autolabel_amd sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the HIP runtime initialises;
    this is the regression test of that setting: torch kernels and library kernels interleaved with the replays."""
    import ctypes as C
    from autolabel_amd import hip as H, synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    assert os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'
    scene = synthetic.make_cube_scene(n_frames=8)
    frames = DeviceFrames.from_scene(scene, 'cuda')
    model = make_model(D=64, C_=scene['n_classes'], bound=6.0, grid_scale=1.0)
    eng = TrainEngine(model._ensure_device(), num_steps=64, upsample_steps=64)
    batch = frames.alloc_batch(2048)
    g = eng.graphed(frames, batch, 1, 2, warmup=2)
    a, b = torch.zeros(1 << 20, device='cuda'), torch.zeros(1 << 20, dtype=torch.float16, device='cuda')
    for i in range(60):
        g()
        if i % 5 == 4:
            torch.zeros(4, device='cuda').fill_(1.0)                       # torch elementwise kernel
            eng.lr = 5e-3 * 0.99 ** i                                      # device word the Adam kernel reads
            H.call('aln_cast_f16', H.ptr(a), H.ptr(b), a.numel(), H.stream())  # a kernel of this library
    torch.cuda.synchronize()
    assert torch.isfinite(eng.terms).all() and torch.isfinite(model._P.flat).all()
    assert int(eng.state_i[0].item()) == g.steps
