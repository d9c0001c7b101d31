"""Training loops (interface of autolabel/trainer.py) + the ``Trainer`` base the reference inherits from the
torch-ngp fork (``torch_ngp.nerf.utils.Trainer``, not in the reference tree -- restated here from its use in
scripts/train.py:80-95 and autolabel/backend.py:29-76,157-164).

Two execution modes, same semantics:

* generic (``fused=False``): the reference's loop verbatim in structure -- ``model.render`` -> torch loss -> GradScaler
  backward -> ``optimizer.step`` (autolabel/trainer.py:39-49); ``render`` is one autograd node backed by the HIP kernels.
* fused (default when the optimizer is the 2-group Adam of scripts/train.py:50-63): ``engine.TrainEngine`` runs forward,
  the on-device loss, backward and a fused Adam (+ loss scaling, overflow skip) with no host synchronisation, and one
  RCCL all-reduce of the flat gradient buffer when ``world_size > 1``.
"""
import glob
import math
import os
import time

import numpy as np
import torch
from torch import optim
from torch.nn import functional as F

DEPTH_EPSILON = 0.01

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = lambda x, **kw: x


class ExponentialMovingAverage:
    """torch_ema.ExponentialMovingAverage subset used by the fork's Trainer (update / store / copy_to / restore)."""

    def __init__(self, parameters, decay):
        self.decay, self.num_updates = decay, 0
        self.params = [p for p in parameters if p.requires_grad]
        self.shadow = [p.detach().clone() for p in self.params]
        self.backup = None

    def update(self):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                s.sub_((1.0 - d) * (s - p.detach()))

    def store(self):
        self.backup = [p.detach().clone() for p in self.params]

    def copy_to(self):
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                p.copy_(s)

    def restore(self):
        with torch.no_grad():
            for b, p in zip(self.backup, self.params):
                p.copy_(b)
        self.backup = None

    def state_dict(self):
        return {'decay': self.decay, 'num_updates': self.num_updates, 'shadow_params': self.shadow}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = sd['decay'], sd['num_updates']
        for s, v in zip(self.shadow, sd['shadow_params']):
            s.copy_(v)


class Trainer:

    def __init__(self, name, opt, model, criterion=None, optimizer=None, ema_decay=None, lr_scheduler=None, metrics=[],
                 local_rank=0, world_size=1, device=None, mute=False, fp16=False, eval_interval=1, max_keep_ckpt=2,
                 workspace='workspace', best_mode='min', use_loss_as_metric=True, report_metric_at_train=False,
                 use_checkpoint='latest', use_tensorboardX=False, scheduler_update_every_step=False, fused=None,
                 process_group=None, use_graph=True, device_data='auto', use_graph_dp=False, shard_optimizer=False, dp_level_group=4):
        self.name, self.opt, self.mute, self.metrics = name, opt, mute, metrics
        self.dp_level_group = int(dp_level_group)   # data parallel: hash-grid levels per scatter launch / gradient bucket (TrainEngine.level_groups)
        self.local_rank, self.world_size, self.workspace = local_rank, world_size, workspace
        self.ema_decay, self.fp16, self.best_mode = ema_decay, fp16, best_mode
        self.use_loss_as_metric, self.report_metric_at_train = use_loss_as_metric, report_metric_at_train
        self.max_keep_ckpt, self.eval_interval, self.use_checkpoint = max_keep_ckpt, eval_interval, use_checkpoint
        self.use_tensorboardX = use_tensorboardX
        self.scheduler_update_every_step = scheduler_update_every_step
        self.device = device if device is not None else torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
        self.time_stamp = time.strftime('%Y-%m-%d_%H-%M-%S')
        self.process_group = process_group
        self.shard_optimizer = bool(shard_optimizer) and world_size > 1   # TrainEngine(shard_optimizer): 1 / world of the table's Adam per rank
        self.use_graph = bool(use_graph)   # device-resident loaders: replay the whole step from a hipGraph (engine.GraphedStep)
        # data parallel: capture the step WITH its collectives (RCCL process groups are capturable, gloo is not); opt-in until it has
        # run on a multi-GPU node -- a failed capture falls back to launch-by-launch steps with a warning
        self.use_graph_dp = bool(use_graph_dp)
        # 'auto': a reference-style host loader (scripts/train.py:65-68: DataLoader over the dataset, one worker) is replaced by
        # device-resident frames + HIP batch assembly whenever the frames fit in HBM -- the reference's CLI has no switch for it, so the
        # drop-in route must not depend on one.  False keeps the host loader; True insists (raises when the frames do not fit).
        self.device_data = device_data
        self._resident = None
        model.to(self.device)
        self.model = model
        self.criterion = criterion if criterion is not None else torch.nn.MSELoss(reduction='none')
        self.optimizer = optimizer(self.model) if optimizer is not None else optim.Adam(self.model.parameters(), lr=0.001)
        self.optimizers = [self.optimizer]
        sched = lr_scheduler(self.optimizer) if lr_scheduler is not None else optim.lr_scheduler.LambdaLR(self.optimizer, lambda e: 1)
        self.lr_scheduler = sched
        self.lr_schedulers = [sched]
        self.ema = ExponentialMovingAverage(self.model.parameters(), decay=ema_decay) if ema_decay is not None else None
        self.scaler = torch.amp.GradScaler('cuda', enabled=self.fp16 and torch.cuda.is_available())
        self.epoch, self.global_step, self.local_step = 0, 0, 0
        self.stats = {'loss': [], 'valid_loss': [], 'results': [], 'checkpoints': [], 'best_result': None}
        self.engine = None
        self.fused = self._can_fuse() if fused is None else fused
        self.ckpt_path = os.path.join(self.workspace, 'checkpoints') if self.workspace is not None else None
        self.best_path = f'{self.ckpt_path}/{self.name}.pth' if self.ckpt_path else None
        if self.workspace is not None:
            os.makedirs(self.ckpt_path, exist_ok=True)
            if self.use_checkpoint == 'latest':
                self.load_checkpoint()
            elif self.use_checkpoint not in ('scratch', None):
                self.load_checkpoint(self.use_checkpoint)

    def log(self, *args, **kwargs):
        if self.local_rank == 0 and not self.mute:
            print(*args, **kwargs)

    # ------------------------------------------------------------------ fused path
    def _can_fuse(self):
        from .models import ALNetwork
        o = self.optimizer
        if not isinstance(self.model, ALNetwork) or not torch.cuda.is_available() or type(o) is not optim.Adam:
            return False
        if not isinstance(self.criterion, torch.nn.MSELoss):
            return False
        return len({g['lr'] for g in o.param_groups}) == 1 and len({g['betas'] for g in o.param_groups}) == 1

    def _engine(self):
        """The fused engine bound to the model's CURRENT device buffers.  ``model.to()/cuda()`` re-binds the parameters to a
        new flat buffer and ``load_state_dict`` / EMA ``copy_to`` rewrite the fp32 masters in place: ``_ensure_device`` is asked
        every time (it refreshes the fp16 shadows when a parameter version changed), a new pipeline gets a new engine that
        inherits the optimizer state, and a checkpoint's engine state is applied as soon as there is an engine to take it."""
        pipe = self.model._ensure_device()
        if self.engine is None or self.engine.pipe is not pipe:
            from .engine import TrainEngine
            g0 = self.optimizer.param_groups[0]
            wd = max(g.get('weight_decay', 0.0) for g in self.optimizer.param_groups)
            opt = self.opt
            old = self.engine
            self.engine = TrainEngine(pipe, lr=g0['lr'], betas=g0['betas'], eps=g0['eps'], weight_decay_net=wd,
                                      rgb_weight=opt.rgb_weight, depth_weight=opt.depth_weight,
                                      semantic_weight=opt.semantic_weight, feature_weight=opt.feature_weight,
                                      feature_loss=getattr(opt, 'feature_loss', False),
                                      num_steps=getattr(opt, 'num_steps', self.model.num_steps_default),
                                      upsample_steps=getattr(opt, 'upsample_steps', self.model.upsample_steps_default),
                                      process_group=self.process_group, shard_optimizer=self.shard_optimizer, level_group=self.dp_level_group,
                                      shard_gather='master' if self.ema is not None else 'table')   # (the EMA reads the fp32 masters every step)
            if old is not None:   # same parameters on a new buffer: the Adam moments / loss scale / step counters carry over
                self.engine.load_state_dict(old.state_dict())
            self._graph = None
        if getattr(self, '_engine_state', None) is not None:
            self.engine.load_state_dict(self._engine_state)
            self._engine_state = None
        self.engine.lr = self.optimizer.param_groups[0]['lr']  # follows the torch scheduler (a device word: captured steps see it)
        return self.engine

    def resident_loader(self, dataloader):
        """The loader `train_iterations` actually draws from: `dataloader` itself, or -- for a host loader over a BaseDataset whose
        frames fit in HBM -- a DeviceLoader over the same frames (same batch size, same epoch length; rank-specific seed and frame
        shard under data parallelism).  Built once per dataset object; `dataset_updated` / `SceneDataset.update_sampler` callers get
        the labels re-uploaded through `refresh_resident_labels`."""
        from .dataset import BaseDataset, DeviceLoader, DynamicDataset
        from . import parallel
        if isinstance(dataloader, DeviceLoader) or not self.fused or self.device_data is False:
            return dataloader
        ds = getattr(dataloader, '_data', None)
        if self._resident is not None and self._resident[0] is ds:
            # the reference's label-edit flows only touch the host dataset (scripts/simulate_user.py:89 update_sampler(), backend.py:155
            # semantic_map_updated()); both end in IndexSampler.update, whose call count says when the device copy is stale
            if getattr(ds.index_sampler, 'version', 0) != self._resident[2]:
                self.refresh_resident_labels()
            return self._resident[1]
        ok = isinstance(ds, BaseDataset) and not isinstance(ds, DynamicDataset) and isinstance(getattr(ds, 'images', None), np.ndarray)
        if ok:
            need = ds.images.nbytes + ds.depths.nbytes + ds.semantics.nbytes + (ds.features.nbytes if ds.features is not None else 0)
            free, _ = torch.cuda.mem_get_info(self.device)
            ok = need <= 0.5 * free
        if not ok:
            if self.device_data is True:
                raise RuntimeError('device_data=True: the dataset cannot be made device-resident (lazy / online frames, or larger than half of the free HBM)')
            return dataloader
        try:
            length = len(dataloader)
        except TypeError:
            length = 1000
        loader = DeviceLoader(ds.device_frames(self.device), ds.batch_size, length, seed=parallel.rank_seed(0, self.local_rank),
                              frame_range=parallel.frame_shard(ds.n_examples, self.local_rank, self.world_size))
        loader._data = ds
        self._resident = (ds, loader, getattr(ds.index_sampler, 'version', 0))
        self.log(f'[INFO] {ds.n_examples} frames resident in HBM ({need / 2 ** 20:.0f} MB): batches are assembled on the device')
        return loader

    def refresh_resident_labels(self):
        """Re-upload the semantic maps and the class index after the labels changed on the host (semantic_map_updated)."""
        if self._resident is not None:
            ds, loader = self._resident[:2]
            fr = loader.frames
            fr.semantics.copy_(torch.as_tensor(ds.semantics).reshape(fr.semantics.shape))
            fr.set_class_index(ds.semantics)
            self._resident = (ds, loader, getattr(ds.index_sampler, 'version', 0))
            self._graph = None    # the captured step holds the old class-index pointers and class count by value

    def _to_device_batch(self, data):
        as_t = lambda v, dt: torch.as_tensor(v).to(self.device, dtype=dt, non_blocking=True).contiguous()
        b = {k: as_t(data[k], torch.float32) for k in ['rays_o', 'rays_d', 'direction_norms', 'pixels', 'depth']}
        b['semantic'] = as_t(data['semantic'], torch.int32)
        if 'features' in data and getattr(self.opt, 'feature_loss', False):
            b['features'] = as_t(data['features'], torch.float32)
        return b

    # ------------------------------------------------------------------ checkpoints (layout: <workspace>/checkpoints/*.pth)
    def save_checkpoint(self, name=None, full=True, best=False):
        if name is None:
            name = f'{self.name}_ep{self.epoch:04d}'
        state = {'epoch': self.epoch, 'global_step': self.global_step, 'stats': self.stats}
        sharded = self.engine is not None and self.engine.shard is not None
        if sharded:   # EVERY rank calls save_checkpoint then: gathering the masters and the Adam moments is a collective; rank 0 writes
            self.engine.sync_master()
        if full:
            state['optimizer'] = self.optimizer.state_dict()
            state['lr_scheduler'] = self.lr_scheduler.state_dict()
            state['scaler'] = self.scaler.state_dict()
            if self.ema is not None:
                state['ema'] = self.ema.state_dict()
            if self.engine is not None:
                state['engine'] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.engine.state_dict().items()}
        if best and self.ema is not None:
            self.ema.store(); self.ema.copy_to()
        state['model'] = self.model.state_dict()
        if best and self.ema is not None:
            self.ema.restore()
        if sharded and self.local_rank != 0:
            return
        path = self.best_path if best else f'{self.ckpt_path}/{name}.pth'
        if not best:
            self.stats['checkpoints'].append(path)
            while len(self.stats['checkpoints']) > self.max_keep_ckpt:
                old = self.stats['checkpoints'].pop(0)
                if os.path.exists(old):
                    os.remove(old)
        torch.save(state, path)

    def load_checkpoint(self, checkpoint=None, model_only=False):
        if checkpoint is None:
            found = sorted(glob.glob(f'{self.ckpt_path}/{self.name}_ep*.pth'))
            if not found:
                self.log('[INFO] No checkpoint found, model randomly initialized.')
                return
            checkpoint = found[-1]
        sd = torch.load(checkpoint, map_location=self.device, weights_only=False)
        if 'model' not in sd:
            self.model.load_state_dict(sd)
            return
        self.model.load_state_dict(sd['model'], strict=False)
        if model_only:
            return
        self.stats = sd.get('stats', self.stats)
        self.epoch, self.global_step = sd.get('epoch', 0), sd.get('global_step', 0)
        for key, obj in [('optimizer', self.optimizer), ('lr_scheduler', self.lr_scheduler), ('scaler', self.scaler), ('ema', self.ema)]:
            if obj is not None and key in sd:
                try:
                    obj.load_state_dict(sd[key])
                except Exception as e:  # a stale optimizer state must not block resuming the weights
                    self.log(f'[WARN] could not load {key}: {e}')
        self._engine_state = sd.get('engine')

    def evaluate(self, loader, name=None):
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for data in loader:
                *_, loss = self.eval_step(data)
                total += float(loss)
                n += 1
        avg = total / max(n, 1)
        self.stats['valid_loss'].append(avg)
        self.log(f'++> Evaluate epoch {self.epoch}: loss {avg:.6f}')
        return avg


class SimpleTrainer(Trainer):

    def train(self, dataloader, epochs):
        if self.model.cuda_ray:
            self.model.mark_untrained_grid(dataloader._data.poses, dataloader._data.intrinsics)
        for _ in range(epochs):
            self.train_iterations(dataloader, 1000)
            self.epoch += 1

    def train_iterations(self, dataloader, iterations):
        """`iterations` optimisation steps, then ONE ema update and ONE scheduler step (autolabel/trainer.py:32-52)."""
        self.model.train()
        if self.model.cuda_ray and getattr(dataloader, '_data', None) is not None:   # autolabel/trainer.py:34-36
            self.model.mark_untrained_grid(dataloader._data.poses, dataloader._data.intrinsics)
        from .dataset import DeviceLoader
        dataloader = self.resident_loader(dataloader)
        if self.fused and isinstance(dataloader, DeviceLoader) and self.use_graph and (self.world_size == 1 or self.use_graph_dp):
            loss = self._graphed_iterations(dataloader, iterations)
            if loss is not None:
                if self.ema is not None:
                    self.ema.update()
                self._step_scheduler(loss)
                return
            # fallen back to launch-by-launch steps: the warm-up steps of the failed capture count towards this call's iterations
            iterations = max(iterations - getattr(self, '_warmup_ran', 0), 0)
            self._warmup_ran = 0
        iterator = iter(dataloader)
        bar = tqdm(range(iterations), desc='Loss: N/A', disable=self.mute or self.local_rank != 0)
        loss = None
        for it in bar:
            data = next(iterator)
            if self.fused:
                loss = self.fused_step(data)
                if it % 100 == 99 and hasattr(bar, 'set_description'):
                    bar.set_description(f'Loss: {float(self.engine.terms[4]):.04f}')  # the only host sync, every 100 steps
            else:
                if self.model.cuda_ray and self.global_step % 16 == 0:   # upstream torch-ngp's refresh cadence (own spec)
                    self.model.update_extra_state()
                for opt in self.optimizers:
                    opt.zero_grad()
                with torch.autocast('cuda', enabled=self.fp16):
                    _, _, loss = self.train_step(data)
                self.scaler.scale(loss).backward()
                for opt in self.optimizers:
                    self.scaler.step(opt)
                self.scaler.update()
                if hasattr(bar, 'set_description'):
                    bar.set_description(f'Loss: {loss:.04f}')
            self.global_step += 1
        if self.ema is not None:
            self.ema.update()
        self._step_scheduler(loss)

    def fused_step(self, data):
        eng = self._engine()
        batch = data if torch.is_tensor(data['rays_o']) and data['rays_o'].is_cuda and data['semantic'].dtype == torch.int32 \
            else self._to_device_batch(data)
        eng.step(batch, seed=self.model._seed, step=self.global_step)
        self.optimizer._opt_called = True   # the fused Adam stepped: torch's scheduler must not warn about its own optimizer
        return eng.terms[4]

    def _graphed_iterations(self, loader, iterations):
        """Device-resident data (dataset.DeviceLoader): ray generation + the whole optimisation step replay from ONE hipGraph
        (engine.GraphedStep); the capture is redone when the engine or the loader changes (not for the learning rate: a device word)."""
        from .engine import GraphUnsafe
        eng = self._engine()
        cur = getattr(self, '_graph', None)
        # the capture holds raw pointers into the engine's workspace and the loader's batch: it is reused only for the same loader
        # and engine OBJECTS (references are kept, so an id cannot be recycled) and only while no buffer has been reallocated
        if cur is None or cur[0] is not loader or cur[1] is not eng or not cur[2].valid():
            if iterations <= 0:
                return eng.terms[4]
            err = g = None
            unsafe = False
            try:
                g = eng.graphed(loader.frames, loader.batch, loader.seed, self.model._seed, frame_range=loader.frame_range,
                                first_step=self.global_step, warmup=1)
            except GraphUnsafe as e:
                if not getattr(self, '_graph_warned', False):
                    self.log(f'[WARN] hipGraph replay disabled, stepping launch by launch: {e}')
                    self._graph_warned = True
                unsafe = True
                err = e
            except Exception as e:
                if self.world_size == 1:
                    raise
                err = e
            # eager warm-up steps of a capture that failed AFTER them did run (collectives included): every rank advances by the steps it
            # executed, whether or not its capture succeeded, so the ranks keep one step numbering
            ran = g.steps if g is not None else int(getattr(err, 'warmup_steps', 0))
            if self.world_size > 1:
                # the fallback is a GROUP decision: a rank stepping launch by launch beside ranks that replay captured collectives
                # would issue another collective sequence and hang the job instead of degrading it
                ok = torch.tensor([0.0 if err is not None else 1.0], device=self.device)
                torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
                if float(ok.item()) == 0.0:
                    why = f'{type(err).__name__}: {err}' if err is not None else 'another rank could not capture it'
                    self.log(f'[WARN] the data-parallel step could not be captured ({why}); every rank steps launch by launch')
                    self.use_graph_dp = False      # (decided by the group, after the all-reduce: no rank switches on its own)
                    self.global_step += ran
                    loader.step += ran
                    self._warmup_ran = ran
                    return None
            elif err is not None:
                if unsafe:
                    self.use_graph = False
                self.global_step += ran
                loader.step += ran
                self._warmup_ran = ran
                return None
            self.global_step += g.steps
            loader.step += g.steps
            self._graph = cur = (loader, eng, g)
            iterations -= g.steps
        g = cur[2]
        for _ in range(max(iterations, 0)):
            g()
        self.global_step += max(iterations, 0)
        loader.step += max(iterations, 0)
        self.optimizer._opt_called = True
        return eng.terms[4]

    def train_step(self, data):
        """4-term loss of autolabel/trainer.py:54-94 on top of model.render (generic autograd path)."""
        dev = self.device
        t = lambda k: torch.as_tensor(data[k]).to(dev)
        rays_o, rays_d, direction_norms = t('rays_o').float(), t('rays_d').float(), t('direction_norms').float()
        gt_rgb, gt_depth, gt_semantic = t('pixels').float(), t('depth').float(), t('semantic').long()
        has_semantic = gt_semantic >= 0
        use_semantic_loss = has_semantic.sum() > 0
        outputs = self.model.render(rays_o, rays_d, direction_norms, staged=False, bg_color=None, perturb=True, **vars(self.opt))
        pred_rgb = outputs['image']
        loss = self.opt.rgb_weight * self.criterion(pred_rgb, gt_rgb).mean()
        has_depth = gt_depth > DEPTH_EPSILON
        if has_depth.any():  # SPEC: an empty set contributes 0 (torch's empty mean would be NaN)
            loss = loss + self.opt.depth_weight * torch.abs(outputs['depth'][has_depth] - gt_depth[has_depth]).mean()
        if getattr(self.opt, 'feature_loss', False):
            gt_features = t('features').float()
            loss = loss + self.opt.feature_weight * F.l1_loss(outputs['semantic_features'][:, :gt_features.shape[1]], gt_features)
        if use_semantic_loss.item():
            loss = loss + self.opt.semantic_weight * F.cross_entropy(outputs['semantic'][has_semantic, :], gt_semantic[has_semantic])
        return pred_rgb, gt_rgb, loss

    def test_step(self, data):
        dev = self.device
        t = lambda k: torch.as_tensor(data[k]).to(dev)
        H, W = data['H'], data['W']
        outputs = self.model.render(t('rays_o'), t('rays_d'), t('direction_norms'), staged=True, perturb=False, **vars(self.opt))
        pred_semantic = outputs['semantic']
        C = pred_semantic.shape[-1]
        return (outputs['image'].reshape(-1, H, W, 3), outputs['depth'].reshape(-1, H, W), pred_semantic.reshape(-1, H, W, C),
                outputs['semantic_features'])

    def eval_step(self, data):
        dev = self.device
        t = lambda k: torch.as_tensor(data[k]).to(dev)
        gt_rgb, gt_depth, gt_semantic = t('pixels').float(), t('depth').float(), t('semantic').long()
        H, W, _ = gt_rgb.shape
        outputs = self.model.render(t('rays_o'), t('rays_d'), t('direction_norms'), staged=True, bg_color=None, perturb=False,
                                    **vars(self.opt))
        pred_rgb, pred_depth = outputs['image'].reshape(H, W, 3), outputs['depth'].reshape(H, W)
        pred_semantic = outputs['semantic'].reshape(H, W, -1)
        loss = self.criterion(pred_rgb, gt_rgb).mean()
        has_depth = gt_depth > DEPTH_EPSILON
        if has_depth.any():
            loss = loss + self.opt.depth_weight * torch.abs(pred_depth[has_depth] - gt_depth[has_depth]).mean()
        has_semantic = gt_semantic >= 0
        if has_semantic.sum().item() > 0:
            loss = loss + self.opt.semantic_weight * F.cross_entropy(pred_semantic[has_semantic, :], gt_semantic[has_semantic])
        return pred_rgb[None], pred_depth[None], pred_semantic[None], gt_rgb[None], loss

    def _step_scheduler(self, loss):
        for s in self.lr_schedulers:
            if isinstance(s, optim.lr_scheduler.ReduceLROnPlateau):
                s.step(loss)
            else:
                s.step()


class InteractiveTrainer(SimpleTrainer):
    """GUI / ROS loop: one step at a time, ema + scheduler every 100 steps (autolabel/trainer.py:163-218)."""

    def __init__(self, *args, **kwargs):
        lr_scheduler = kwargs['lr_scheduler']
        kwargs['lr_scheduler'] = None
        super().__init__(*args, **kwargs)
        self.loader = None
        self.lr_scheduler = lr_scheduler(self.optimizer)
        self.lr_schedulers = [self.lr_scheduler]

    def init(self, loader):
        self.model.train()
        self.iterator = iter(loader)
        self.step = 0
        self.model.mark_untrained_grid(loader._data.poses, loader._data.intrinsics)

    def train(self, loader):
        while True:
            self.model.train()
            self.train_one_epoch(loader)

    def train_one_epoch(self, loader):
        self.train_iterations(loader, 1000)

    def take_step(self):
        data = next(self.iterator)
        if self.fused:
            loss = self.fused_step(data)
        else:
            if self.model.cuda_ray and self.global_step % 16 == 0:   # upstream torch-ngp's cadence (own spec); the hook site is autolabel/trainer.py:176
                self.model.update_extra_state()
            self.optimizer.zero_grad()
            with torch.autocast('cuda', enabled=self.fp16):
                _, _, loss = self.train_step(data)
            self.scaler.scale(loss).backward()
            self.scaler.step(self.optimizer)
            self.scaler.update()
        self.step += 1
        self.global_step += 1
        if self.step % 100 == 0:
            if self.ema is not None:
                self.ema.update()
            self._step_scheduler(loss)
        return loss

    def dataset_updated(self, loader):
        self.loader = loader
        self.refresh_resident_labels()
