"""Fused training step: render forward -> on-device loss -> backward -> (RCCL grad all-reduce) -> fused Adam.

This is the MI355X-native replacement of the reference's hot loop (autolabel/trainer.py:39-49: zero_grad, autocast
forward, GradScaler-scaled backward, per-optimizer step, scaler.update) with no host synchronisation: the loss
scale, the overflow flag, the optimizer step counter and the loss terms all live on the device.
"""
import ctypes as C

import os

import torch

from . import hip as H

ADAM_DEFAULTS = dict(lr=5e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay_net=1e-6)  # scripts/train.py:50-63
SCALER_DEFAULTS = dict(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)  # torch GradScaler


class TrainEngine:
    def __init__(self, pipe, lr=5e-3, betas=(0.9, 0.99), eps=1e-15, weight_decay_net=1e-6, rgb_weight=1.0, depth_weight=0.1,
                 semantic_weight=1.0, feature_weight=0.5, feature_loss=False, num_steps=128, upsample_steps=128,
                 scaler=None, process_group=None, overlap_comm=True, grad_payload='f16', fuse_grid_adam=True, shard_optimizer=False,
                 shard_gather='table', exchange_at_world_1=False, direct_wire=True, level_group=4):
        self.pipe, self.P, self.L = pipe, pipe.P, pipe.L
        # hash-grid levels per scatter launch / gradient bucket of the overlapped exchange (level_groups).  Resolved ONCE, here: an explicit
        # argument wins; ALN_LEVEL_GROUP is only the default for level_group=None (re-reading the environment on every call let the
        # variable override the argument, and ranks with different environments would issue different collective sequences)
        if level_group is None:
            level_group = os.environ.get('ALN_LEVEL_GROUP', 4)
        self.level_group = max(1, int(level_group))
        dv = self.P.device
        # the step's intermediates live in a workspace of the engine's own: a render through the same pipeline (pipe.ws) between
        # two steps cannot move the buffers a captured step points into
        from .pipeline import Workspace
        self.ws = Workspace(dv)
        self._lr, self.betas, self.eps, self.wd = float(lr), betas, float(eps), float(weight_decay_net)
        self.weights = (float(rgb_weight), float(depth_weight), float(semantic_weight), float(feature_weight))
        self.feature_loss = feature_loss
        # LSeg-style training (scripts/ros/node.py:166-176: feature_dim 512, semantic_weight 0.0): no loss reaches the class logits,
        # so the wide heads take the linear path (pipeline.forward: sem_linear)
        self.sem_linear = bool(self.L.sem_wide and feature_loss and float(semantic_weight) == 0.0)
        self.S1, self.S2 = int(num_steps), int(upsample_steps)
        sc = dict(SCALER_DEFAULTS, **(scaler or {}))
        self.scaler_cfg = sc
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if self.world > 1 else 0
        # `dp`: the step takes the data-parallel route (gradient through P.grad, collectives, separate optimizer pass).  With
        # exchange_at_world_1 a process group of ONE rank takes it too: every collective is then really issued (RCCL communicator
        # set-up, AVG, in-place reduce-scatter / all-gather, stream ordering, hipGraph capture with collectives) although it moves
        # nothing -- how the RCCL code is exercised on a box with one GPU (tests/test_gpu_rccl.py)
        self.dp = self.world > 1 or (bool(exchange_at_world_1) and process_group is not None)
        # Sharded optimizer (data parallel only): every rank owns 1 / world of every hash-grid gradient bucket -- reduce-scatter
        # instead of all-reduce, Adam on the owned slices with moments allocated for those slices only, all-gather of the updated
        # fp16 table.  The fp32 masters of the other ranks' slices go stale in P.flat until sync_master() (checkpoints).
        # shard_gather='master': the fp32 masters are all-gathered every step instead of the fp16 table (twice the bytes) -- for a
        # caller that reads the masters between steps (the trainer's EMA)
        assert shard_gather in ('table', 'master')
        self.shard_gather = shard_gather
        self.shard = None
        if shard_optimizer and self.dp and self.L.n_grid > 0:
            from .parallel import shard_range
            F, g = int(self.L.enc.grid.n_features), self.L.enc.grid
            nl = int(g.n_levels)
            buckets = sorted((int(g.offset[lo]) * F, int(g.offset[hi]) * F if hi < nl else self.L.n_grid) for lo, hi in self.level_groups())
            assert self.L.n_grid % 4 == 0 and len(buckets) <= 8, 'sharded optimizer: table length must be a multiple of 4, at most 8 level groups'
            own = [shard_range(a, b, self.rank, self.world)[:2] for a, b in buckets]
            self.shard = dict(buckets=buckets, own=own, n_own=sum(hi - lo for lo, hi in own),
                              lo=(C.c_int64 * len(own))(*[lo for lo, _ in own]), hi=(C.c_int64 * len(own))(*[hi for _, hi in own]))
        n = self.L.n_total if self.shard is None else self.shard['n_own'] + self.L.n_total - self.L.n_grid
        self.m = torch.zeros(n, device=dv)     # (sharded: the owned table slices back to back, then the MLP block)
        self.v = torch.zeros(n, device=dv)
        self.state_i = torch.zeros(16, dtype=torch.int32, device=dv)  # [steps, growth tracker, found_inf, scatter found_inf (DP overlap), per-block steps]
        self.state_f = torch.tensor([sc['init_scale'], float(lr), 0, 0], dtype=torch.float32, device=dv)  # [loss scale, lr]
        self.consts = torch.zeros(24, device=dv)
        # parameter blocks = the reference's parameter tensors (torch skips tensors whose grad is None)
        L = self.L
        ends, kinds = ([L.n_grid] if L.n_grid else []), ([0] if L.n_grid else [])
        for k, kind in [('sigma', 0), ('color', 0), ('semf', 2), ('semo', 1)]:
            ends.append(L.offsets[k] + L.nets[k].n_params)
            kinds.append(kind)
        self._blk_end = (C.c_int64 * len(ends))(*ends)
        self._blk_kind = (C.c_int32 * len(kinds))(*kinds)
        self.counts = torch.zeros(4, dtype=torch.int32, device=dv)   # rays with valid depth, labelled rays, the loss kernel's ticket
        self._terms = torch.zeros(int(H.lib().aln_loss_terms_floats()), device=dv)
        self.terms = self._terms[:5]  # rgb, depth, feature, semantic, total (last step); behind them: scratch of aln_loss_fwd_bwd
        pipe.found_inf = self.state_i[2:3]
        # data parallel: the gradient all-reduce runs in buckets on a side stream while the hash-grid scatter is still
        # working on the remaining levels (the scatter is the last and longest kernel of the backward pass)
        self.overlap_comm = self.dp and bool(overlap_comm)   # False: one collective after the backward pass
        # single GPU: the scatter's second phase holds the table's exact gradient sums in LDS and takes the Adam step for the table
        # itself (step()); with more ranks the gradient has to be averaged first, so it goes through P.grad and aln_adam_step
        self.fuse_grid_adam = bool(fuse_grid_adam) and not self.dp and L.n_grid > 0 and int(L.enc.grid.n_features) == 2
        self.grad_payload = grad_payload     # 'f16': the hash-grid gradient crosses the wire as fp16 (parallel.allreduce_bucket)
        self.direct_wire = bool(direct_wire)  # ... written by the scatter itself instead of an fp32 table gradient + a packing pass (_wire_direct)
        self._comm = torch.cuda.Stream(device=dv) if self.dp else None
        self._g = {}
        # occupancy-grid marching (pipe.occ set by enable_marching / ALNetwork(cuda_ray=True)): S1 rows per ray inside occupied
        # cells, no importance pass; the density grid is refreshed every occ.update_interval steps (autolabel/trainer.py:34-36)
        self.march = pipe.occ is not None
        if self.march:
            self.S1, self.S2 = pipe.occ.samples, 0
        self._calls = 0
        if self._wire_direct():
            self._wire_full()      # (allocated here, not inside a step that may be the one a hipGraph capture records)

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        """The learning rate lives in device memory (state_f[1], read by the Adam kernel): a captured step follows the scheduler
        without being re-captured."""
        value = float(value)
        if value != self._lr:
            self._lr = value
            # a copy (DMA), not a kernel: harmless between graph replays even where autolabel_amd/__init__.py's workaround for the
            # ROCm 7.2 packet-capture hazard is overridden
            self.state_f[1:2].copy_(torch.tensor([value], dtype=torch.float32))

    def _gbuf(self, name, shape):
        t = self._g.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._g[name] = torch.empty(shape, dtype=torch.float32, device=self.P.device)
        return t

    def forward_backward(self, batch, seed, step, noise=None, u=None, step_dev=None, grid_adam=False):
        """Forward, loss, backward: every gradient in P.grad afterwards -- except, with ``grid_adam`` (step() on one GPU), the hash
        table's: its optimizer step has then been taken inside the scatter and optimizer_step(skip_grid=True) must follow."""
        L, pipe = self.L, self.pipe
        N = batch['rays_o'].shape[0]
        out, ctx = pipe.forward(batch['rays_o'], batch['rays_d'], batch['direction_norms'].reshape(-1), self.S1, self.S2, True,
                                train=True, seed=seed, step=step, noise=noise, u=u, step_dev=step_dev, march=self.march, ws=self.ws,
                                sem_linear=self.sem_linear)
        gt_feat = batch.get('features') if self.feature_loss else None
        Cf = gt_feat.shape[1] if gt_feat is not None else 0
        g_image, g_depth = self._gbuf('g_image', (N, 3)), self._gbuf('g_depth', (N,))
        g_sem, g_feat = self._gbuf('g_sem', (N, L.C)), self._gbuf('g_feat', (N, L.D))
        w = self.weights
        H.call('aln_loss_fwd_bwd', H.ptr(out['image']), H.ptr(out['depth']), H.ptr(out['semantic']),
               H.ptr(out['semantic_features']), H.ptr(batch['pixels']), H.ptr(batch['depth']), H.ptr(batch['semantic']),
               H.ptr(gt_feat), N, L.C, L.D, Cf, w[0], w[1], w[2], w[3] if gt_feat is not None else 0.0, H.ptr(self.state_f),
               H.ptr(self.counts), H.ptr(g_image), H.ptr(g_depth), H.ptr(g_sem), H.ptr(g_feat), H.ptr(self.terms), H.stream())
        # fp16 on the wire, replicated optimizer: the scatter writes the exchange's payload itself (no fp32 table gradient, no packing pass)
        wire = self._wire_full() if self._wire_direct() else None
        kw = dict(grid_wire=wire, wire_mul=1.0 / self.world) if wire is not None else {}
        if self.dp and self.overlap_comm:
            self._reduced = True
            # the scatter raises its own flag word: the MLP bucket's tail overwrites state_i[2] on the communication stream while
            # the scatter is still running on the compute stream, so a flag stored there in between could be lost
            pipe.backward(ctx, g_image, g_depth, g_sem, g_feat, level_groups=self.level_groups(), on_grad_ready=self._bucket_ready,
                          scatter_flag=self.state_i[3:4], **kw)
        else:
            self._reduced = False
            pipe.backward(ctx, g_image, g_depth, g_sem, g_feat, grid_adam=self._adam_fuse() if grid_adam else None, **kw)
        return out

    def _wire_direct(self):
        """True when the hash-grid scatter writes the fp16 payload of the exchange itself (aln_encode_bwd_binned_wire): data parallelism,
        fp16 on the wire, replicated optimizer.  (The sharded optimizer stages padded shards and clears its source: parallel.reduce_scatter_bucket.)"""
        return bool(self.dp and self.grad_payload == 'f16' and self.shard is None and self.L.n_grid > 0 and self.direct_wire)

    def averaged_gradient(self):
        """The flat gradient after all_reduce_grads() as fp32 (a copy): P.grad, with the table's part taken from the fp16 payload when the
        scatter wrote that itself (then P.grad holds no table gradient at any time)."""
        g = self.P.grad[:self.L.n_total].clone()
        if self._wire_direct():
            g[:self.L.n_grid] = self._wire_full().float()
        return g

    def _wire_full(self):
        """fp16 payload of the whole table's gradient (element i = flat gradient element i), written by the scatter, reduced in place."""
        t = self._g.get('wire_full')
        if t is None:
            t = self._g['wire_full'] = torch.zeros(self.L.n_grid, dtype=torch.float16, device=self.P.device)
        return t

    def _adam_fuse(self):
        P, L = self.P, self.L
        assert not self.dp and self._blk_end[0] == L.n_grid and self._blk_kind[0] == 0
        return H.AlnAdamFuse(P.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), P.table16.data_ptr(), self.state_i.data_ptr(),
                             self.state_f.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps)

    def level_groups(self):
        """Hash-grid levels in scatter order, the groups whose gradient is exchanged while the next group is scattered: the fine levels
        first in groups of ``level_group`` (default 4 -- one level per wave of the scatter kernel), the small coarse levels last, so the
        only all-reduce nothing can hide is the smallest.  Phase 1 of the scatter runs once for all levels; every phase-2 launch beyond the
        first costs ~15 us (world-of-one RCCL legs of bench.py: groups of 4 = five buckets 1.93 ms, groups of 8 = three buckets 1.90): groups of 4 hide
        8.4 MB buckets behind ~150 us of scatter each and expose 1.4 MB; groups of 8 expose the 11 MB of levels 0-7.  Which wins depends on
        what the links deliver, so `bench.py --gpus N` times both (dp_overlap, dp_overlap_g8)."""
        n = int(self.L.enc.grid.n_levels) if self.L.n_grid else 0
        per = self.level_group
        groups, hi = [], n
        while hi > 0:
            lo = max(0, hi - per)
            groups.append((lo, hi))
            hi = lo
        return groups

    def _bucket_ready(self, kind, a, b):
        """P.grad[a:b] is final on the compute stream: average it over the ranks on the communication stream."""
        from .parallel import allreduce_bucket
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ev)
            if kind == 'mlp':   # MLP block + the overflow flag in the tail element (every rank must skip the same steps)
                allreduce_bucket(self.P.grad, a, b, self.pg, found_inf=self.state_i[2:3], tail=self.L.n_total, counts=self.counts, force=True)
            else:
                self._exchange_grid(a, b)
                if a == 0:
                    # last bucket: every scatter group has run (this stream waited for the compute stream's event).  The scatter's
                    # own flag word (state_i[3]: a non-finite record) is MAX-reduced over the ranks and OR-ed into the step's flag
                    # here, on the stream that also wrote the MLP bucket's flag -- ordered after both writers
                    flag = self.state_i[3:4].to(torch.float32)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.pg)
                    self.state_i[2:3] = torch.maximum(self.state_i[2:3], (flag > 0).to(self.state_i.dtype))
                    self.state_i[3:4] = 0

    def _exchange_grid(self, a, b):
        """One hash-grid bucket over the ranks on the current stream: all-reduce, or reduce-scatter under the sharded optimizer."""
        from .parallel import allreduce_bucket, reduce_scatter_bucket, shard_range
        if self.shard is None:
            direct = self._wire_direct()
            allreduce_bucket(self.P.grad, a, b, self.pg, payload=self.grad_payload, scratch=self._wire_full()[a:b] if direct else self._wire(b - a),
                             flag=self.state_i[3:4], force=True, prepacked=direct)
        else:
            S = shard_range(a, b, self.rank, self.world)[2]
            reduce_scatter_bucket(self.P.grad, a, b, self.pg, payload=self.grad_payload, scratch=self._wire((self.world + 1) * S), flag=self.state_i[3:4])

    def _wire(self, n):
        """fp16 staging buffer of the gradient payload (one per engine, grown to the largest bucket)."""
        if self.grad_payload != 'f16':
            return None
        t = self._g.get('wire')
        if t is None or t.numel() < n:
            t = self._g['wire'] = torch.empty(n, dtype=torch.float16, device=self.P.device)
        return t

    def all_reduce_grads(self):
        """Average every gradient over the ranks.  With overlap the buckets are already in flight on the communication
        stream and this only makes the compute stream wait for them; otherwise ONE collective over the flat buffer with
        the overflow flag riding in its tail."""
        if self.dp:
            if self._reduced:
                torch.cuda.current_stream().wait_stream(self._comm)
            elif self.shard is not None:   # the same collectives as the overlapped path, one after the other on this stream
                from .parallel import allreduce_bucket
                allreduce_bucket(self.P.grad, self.L.n_grid, self.L.n_total, self.pg, found_inf=self.state_i[2:3], tail=self.L.n_total, counts=self.counts, force=True)
                for a, b in self.shard['buckets']:
                    self._exchange_grid(a, b)
                flag = self.state_i[3:4].to(torch.float32)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.pg)
                self.state_i[2:3] = torch.maximum(self.state_i[2:3], (flag > 0).to(self.state_i.dtype))
                self.state_i[3:4] = 0
            else:
                from .parallel import allreduce_gradients
                direct = self._wire_direct()
                allreduce_gradients(self.P.grad, self.L.n_total, self.state_i[2:3], self.pg, counts=self.counts, n_grid=self.L.n_grid,
                                    payload=self.grad_payload, scratch=self._wire_full() if direct else self._wire(self.L.n_grid), force=True,
                                    prepacked=direct)

    def optimizer_step(self, step_dev=None, skip_grid=False):
        P, L, sc = self.P, self.L, self.scaler_cfg
        if self.shard is not None:
            sh = self.shard
            H.call('aln_adam_step_ranges', H.ptr(P.flat), H.ptr(P.grad), H.ptr(self.m), H.ptr(self.v), H.ptr(P.table16), L.n_grid, L.n_total,
                   H.ptr(self.state_i), H.ptr(self.state_f), H.ptr(self.consts), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                   sc['growth_factor'], sc['backoff_factor'], int(sc['growth_interval']), len(self._blk_kind), self._blk_end,
                   self._blk_kind, int(bool(self.feature_loss)), len(sh['own']), sh['lo'], sh['hi'], H.ptr(self.counts), H.ptr(step_dev), H.stream())
            from .parallel import allgather_bucket
            if self.shard_gather == 'master':
                self.sync_master()
                P.refresh_shadows(grid=True)
                return
            for a, b in sh['buckets']:     # the updated fp16 table of every owner (a skipped step gathers the unchanged table)
                allgather_bucket(P.table16, a, b, self.pg, scratch=self._g.get('wire'))
            P.masters_stale = True         # (until sync_master: state_dict / shadow refreshes refuse to read the masters)
            P.refresh_shadows(grid=False)
            return
        if self._wire_direct() and not skip_grid:   # the table's averaged gradient is the fp16 payload the exchange left (no fp32 copy of it)
            H.call('aln_adam_step_wire', H.ptr(P.flat), H.ptr(P.grad), H.ptr(self.m), H.ptr(self.v), H.ptr(P.table16), L.n_grid, L.n_total,
                   H.ptr(self.state_i), H.ptr(self.state_f), H.ptr(self.consts), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                   sc['growth_factor'], sc['backoff_factor'], int(sc['growth_interval']), len(self._blk_kind), self._blk_end,
                   self._blk_kind, int(bool(self.feature_loss)), H.ptr(self._wire_full()), H.ptr(self.counts), H.ptr(step_dev), H.stream())
            P.refresh_shadows(grid=False)
            return
        H.call('aln_adam_step', H.ptr(P.flat), H.ptr(P.grad), H.ptr(self.m), H.ptr(self.v), H.ptr(P.table16), L.n_grid, L.n_total,
               H.ptr(self.state_i), H.ptr(self.state_f), H.ptr(self.consts), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
               sc['growth_factor'], sc['backoff_factor'], int(sc['growth_interval']), len(self._blk_kind), self._blk_end,
               self._blk_kind, int(bool(self.feature_loss)), int(bool(skip_grid)), H.ptr(self.counts), H.ptr(step_dev), H.stream())
        P.refresh_shadows(grid=False)

    def maybe_update_grid(self):
        """Density-grid refresh every `update_interval` (16) steps: upstream torch-ngp's cadence, this build's own spec (the
        reference never runs with cuda_ray=True, autolabel/model_utils.py:72, and only carries the mark_untrained_grid hooks)."""
        if self.march and self._calls % self.pipe.occ.update_interval == 0:
            self.pipe.update_density_grid(step=self._calls, ws=self.ws)
        self._calls += 1

    def step(self, batch, seed, step, noise=None, u=None, step_dev=None, grid_update=True):
        if grid_update:
            self.maybe_update_grid()
        fuse = self.fuse_grid_adam
        out = self.forward_backward(batch, seed, step, noise, u, step_dev, grid_adam=fuse)
        self.all_reduce_grads()
        self.optimizer_step(step_dev, skip_grid=fuse)    # (advances the device step counter of a captured step: no separate increment launch)
        return out

    def graphed(self, frames, batch, data_seed, seed, frame_range=None, first_step=0, warmup=3):
        """The whole training step -- device ray generation (dataset.DeviceFrames) + forward + loss + backward + Adam -- captured
        ONCE into a hipGraph; every call of the returned object replays it with the next step number (the step counter is a
        device word the kernels add to their RNG step, so a replay draws fresh pixels / jitter / sample noise).  The launch
        sequence has no host-side decision, allocation or synchronisation, which is what makes it capturable; replaying it
        removes the ~0.7 ms of Python + HIP launch overhead a step costs when issued call by call."""
        def body(step_dev):
            frames.next_train(batch, seed=data_seed, step=first_step, frame_range=frame_range, step_dev=step_dev)
            self.step(batch, seed=seed, step=first_step, step_dev=step_dev, grid_update=False)
        guard = lambda: (self.ws.generation, batch['rays_o'].data_ptr(), getattr(frames, 'version', 0))
        if not self.march:
            return GraphedStep(body, self.P.device, warmup=warmup, guard=guard)
        # marching: every update_interval-th step starts with the density-grid refresh (see maybe_update_grid); that
        # variant of the step is a second captured graph, so the loop never issues a launch of its own between replays
        def body_with_refresh(step_dev):
            self.pipe.update_density_grid(step=first_step, step_dev=step_dev, ws=self.ws)
            body(step_dev)
        return GraphedStep(body, self.P.device, warmup=warmup, alt_body=body_with_refresh, alt_every=self.pipe.occ.update_interval,
                           guard=guard)

    def sync_master(self):
        """Sharded optimizer: all-gather the fp32 masters of the hash table into P.flat on every rank (a collective: every rank
        calls it).  The training step itself only exchanges the fp16 table; checkpoints and EMA copies want the masters."""
        if self.shard is not None:
            from .parallel import allgather_bucket
            for a, b in self.shard['buckets']:
                allgather_bucket(self.P.flat, a, b, self.pg)
            self.P.masters_stale = False

    def _full_moments(self, t):
        """Sharded layout [owned slices | MLP] -> the replicated layout [table | MLP] (a collective under the sharded optimizer)."""
        if self.shard is None:
            return t
        from .parallel import allgather_bucket
        L, sh = self.L, self.shard
        full = torch.zeros(L.n_total, dtype=t.dtype, device=t.device)
        at = 0
        for lo, hi in sh['own']:
            full[lo:hi] = t[at:at + hi - lo]; at += hi - lo
        full[L.n_grid:] = t[at:]
        for a, b in sh['buckets']:
            allgather_bucket(full, a, b, self.pg)
        return full

    def _own_moments(self, full):
        if self.shard is None:
            return full
        return torch.cat([full[lo:hi] for lo, hi in self.shard['own']] + [full[self.L.n_grid:self.L.n_total]])

    # checkpoint payload mirrors torch's {'optimizer', 'scaler'} entries (autolabel/backend.py:157-164); the moments are always
    # stored in the replicated layout, so a checkpoint does not depend on the number of ranks it was written with
    def state_dict(self):
        return {'m': self._full_moments(self.m), 'v': self._full_moments(self.v), 'state_i': self.state_i, 'state_f': self.state_f, 'lr': self.lr}

    def load_state_dict(self, sd):
        dv = self.m.device
        self.m.copy_(self._own_moments(sd['m'].to(dv))); self.v.copy_(self._own_moments(sd['v'].to(dv)))
        self.state_i.copy_(sd['state_i']); self.state_f.copy_(sd['state_f'])
        self._lr = float(sd.get('lr', self._lr))
        self.state_f[1:2].copy_(torch.tensor([self._lr], dtype=torch.float32))


class GraphUnsafe(RuntimeError):
    """hipGraph replay cannot be trusted in this process (see autolabel_amd.graph_replay_is_safe)."""


class GraphedStep:
    """A fixed launch sequence captured into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm).

    ``body(step_dev)`` issues the launches; ``step_dev`` is a device int32[1] holding the number of steps so far, which the
    RNG-consuming kernels add to their step argument.  ``warmup`` eager calls run first so that every lazily allocated
    workspace exists before the capture (allocation is not capturable).  The learning rate is NOT baked in (device word).  ``alt_body`` is a variant of the step (captured as a
    second graph) that replaces it on every ``alt_every``-th step, counted from step 0.  Re-capture (``GraphedStep(...)`` again)
    after anything baked into the launches changes: batch size, loss weights, level groups."""

    def __init__(self, body, device, warmup=3, alt_body=None, alt_every=0, guard=None):
        import autolabel_amd
        ok, why = autolabel_amd.graph_replay_is_safe()
        if not ok:
            raise GraphUnsafe(why)
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.body, self.alt_body, self.alt_every = body, alt_body, int(alt_every)
        self.steps = 0
        self._guard = guard
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2 if alt_body is not None else 1)):
                self._once(self._is_alt())
                self.steps += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self._once(False)
            self.alt_graph = None
            if alt_body is not None:
                self.alt_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.alt_graph, stream=side):
                    self._once(True)
        except Exception as e:
            # the eager warm-up steps above DID run (optimizer steps, collectives included): whoever catches this has to advance its step
            # counters by them, or this rank's step numbering (RNG / data step, scheduler, checkpoints) falls behind its peers'
            e.warmup_steps = self.steps
            raise
        self._captured = guard() if guard is not None else None

    def valid(self):
        """False once a buffer the captured launches point into has been reallocated (Workspace.generation moved): replaying
        would write through stale pointers.  The owner re-captures (SimpleTrainer) or stops."""
        return self._guard is None or self._guard() == self._captured

    def _is_alt(self):
        return self.alt_body is not None and self.steps % self.alt_every == 0

    def _once(self, alt):
        (self.alt_body if alt else self.body)(self.counter)    # the body's last launch (aln_adam_step) increments the counter

    def __call__(self):
        if not self.valid():
            raise RuntimeError('GraphedStep: a workspace buffer of the captured step was reallocated after the capture (a step of another '
                               'batch size ran through the same engine?); capture again')
        (self.alt_graph if self._is_alt() else self.graph).replay()
        self.steps += 1
