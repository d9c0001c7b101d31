"""``NeRFRenderer`` -- base class of ``ALNetwork``.

In the reference this class comes from the ETH-ASL fork of torch-ngp (``torch_ngp.nerf.renderer.NeRFRenderer``, not in the
reference tree; call sites autolabel/trainer.py:64-70,102-107,127-133, scripts/render.py:96-102, scripts/export.py:83-89,
autolabel/evaluation.py:62-66,302-306, scripts/ros/node.py:256-262, scripts/language/pointcloud.py:58-68).  Here its
``render``/``run`` are one fixed sequence of HIP launches (``pipeline.HipPipeline``) wrapped in a single autograd node.

Interface kept: ``render(rays_o, rays_d, direction_norms, staged=False, bg_color=None, perturb=False, num_steps=...,
upsample_steps=..., max_ray_batch=..., **ignored_opt)`` -> dict with ``image, depth, semantic, semantic_features,
depth_variance, coordinates_map, weights_sum``; attributes ``bound, cuda_ray, density_scale, min_near, bg_radius``;
``mark_untrained_grid`` / ``update_extra_state`` (no-ops when ``cuda_ray`` is False, autolabel/model_utils.py:72).
"""
import torch
import torch.nn as nn

from .pipeline import HipPipeline, ModelLayout, Params


class _RenderFn(torch.autograd.Function):
    """forward = HipPipeline.forward, backward = HipPipeline.backward; parameter gradients are returned as views of the
    flat gradient buffer (one tensor per parameter block)."""

    @staticmethod
    def forward(ctx, model, rays_o, rays_d, norms, cfg, *params):
        pipe = model._pipe
        train = any(ctx.needs_input_grad[5:])  # (grad mode is off inside Function.forward)
        out, c = pipe.forward(rays_o, rays_d, norms, cfg['num_steps'], cfg['upsample_steps'], cfg['perturb'], train=train,
                              seed=cfg['seed'], step=cfg['step'], noise=cfg.get('noise'), u=cfg.get('u'), bg=cfg['bg'],
                              march=cfg.get('march', False))
        ctx.model, ctx.c = model, (c if train else None)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None: their heads then get grad None like in the reference
        keys = ['image', 'depth', 'semantic', 'semantic_features', 'depth_variance', 'coordinates_map', 'weights_sum']
        ctx.mark_non_differentiable(out['depth_variance'], out['coordinates_map'], out['weights_sum'])
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, g_image, g_depth, g_sem, g_feat, *_):
        model, c = ctx.model, ctx.c
        if c is None:
            raise RuntimeError('render() was run without gradient tracking')
        pipe, L = model._pipe, model._layout
        N = c['N']
        z = lambda g, shape: (torch.zeros(shape, device=pipe.P.device) if g is None else g.float().contiguous())
        pipe.P.grad.zero_()
        pipe.backward(c, z(g_image, (N, 3)), z(g_depth, (N,)), z(g_sem, (N, L.C)), z(g_feat, (N, L.D)))
        no_grad = set()
        if g_sem is None:
            no_grad.add('semo')
            if g_feat is None:
                no_grad.add('semf')
        grads = []
        for name, p in model._param_blocks():
            a, b = model._block_range(name)
            grads.append(pipe.P.grad[a:b].view_as(p).clone() if p.requires_grad and name not in no_grad else None)
        return (None, None, None, None, None, *grads)


class NeRFRenderer(nn.Module):

    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=10.0, bg_radius=-1,
                 grid_size=128, max_steps=1024, march_samples=96, **kwargs):
        super().__init__()
        self.bound = bound
        self.cuda_ray = cuda_ray
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        # cuda_ray: occupancy-grid marching (csrc/march.hip).  The reference passes False (autolabel/model_utils.py:72); True
        # places `march_samples` rows per ray inside occupied cells instead of num_steps + upsample_steps rows along the whole ray.
        self.grid_size, self.max_steps, self.march_samples = int(grid_size), int(max_steps), int(march_samples)
        if cuda_ray:   # like upstream's density_grid / density_bitfield buffers: part of the checkpoint
            n = self.grid_size ** 3
            self.register_buffer('density_grid', torch.zeros(n, dtype=torch.float32))
            self.register_buffer('density_bitfield', torch.zeros((n + 31) // 32, dtype=torch.int32))
        self._grid_updates = 0
        aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.register_buffer('aabb_train', aabb)
        self.register_buffer('aabb_infer', aabb.clone())
        self._render_calls = 0
        self.num_steps_default, self.upsample_steps_default, self.max_ray_batch_default = 128, 128, 4096

    # --- hooks of the cuda_ray path (autolabel/trainer.py:21-23,34-36,176; no-ops when cuda_ray is False, as in the reference)
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """poses: [F,4,4] camera-to-world in the renderer's frame (dataset.poses = _convert_pose output, whose camera axes are
        x right / y down / z forward); intrinsic: (fx, fy, cx, cy).  Cells no camera sees are excluded for good."""
        if not self.cuda_ray:
            return
        import numpy as np
        T_WC = np.asarray(poses.cpu() if torch.is_tensor(poses) else poses, dtype=np.float64).reshape(-1, 4, 4)
        self._ensure_device().mark_untrained_grid(np.linalg.inv(T_WC).astype(np.float32), intrinsic)

    def update_extra_state(self, decay=0.95, S=128):
        if not self.cuda_ray:
            return
        pipe = self._ensure_device()
        pipe.occ.decay = float(decay)
        pipe.update_density_grid(step=self._grid_updates)
        self._grid_updates += 1

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.density_bitfield.zero_()
        self._grid_updates = 0

    def run(self, rays_o, rays_d, direction_norms, num_steps=None, upsample_steps=None, bg_color=None, perturb=False,
            seed=None, noise=None, u=None, **kwargs):
        """One chunk of rays [N,3] -> per-ray outputs.  bg_color None -> white (upstream torch-ngp)."""
        self._ensure_device()
        N = rays_o.shape[0]
        if self.cuda_ray:   # marching ignores num_steps / upsample_steps (as upstream's run_cuda does)
            num_steps, upsample_steps = self.march_samples, 0
            if self._grid_updates == 0 and int(self._pipe.occ.n_set.item()) == 0:
                self.update_extra_state()   # never rendered through an empty bitfield
        cfg = dict(num_steps=int(num_steps if num_steps is not None else self.num_steps_default),
                   upsample_steps=int(upsample_steps if upsample_steps is not None else self.upsample_steps_default), march=bool(self.cuda_ray),
                   perturb=bool(perturb), seed=int(self._seed if seed is None else seed), step=self._render_calls,
                   bg=1.0 if bg_color is None else float(bg_color), noise=noise, u=u)
        self._render_calls += 1
        f = lambda t: t.reshape(-1, t.shape[-1]).float().contiguous()
        names = ['image', 'depth', 'semantic', 'semantic_features', 'depth_variance', 'coordinates_map', 'weights_sum']
        outs = _RenderFn.apply(self, f(rays_o), f(rays_d), direction_norms.reshape(-1).float().contiguous(), cfg,
                               *[p for _, p in self._param_blocks()])
        return dict(zip(names, outs))

    def render(self, rays_o, rays_d, direction_norms, staged=False, max_ray_batch=None, **kwargs):
        """rays_*: [..., 3]; direction_norms: [..., 1] (or flat).  staged=True chunks rays by max_ray_batch and returns
        outputs reshaped to the ray prefix (H, W for a frame: autolabel/trainer.py:109-114)."""
        prefix = rays_o.shape[:-1]
        ro, rd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        dn = direction_norms.reshape(-1)
        kwargs = {k: v for k, v in kwargs.items() if k in ('num_steps', 'upsample_steps', 'bg_color', 'perturb', 'seed', 'noise', 'u')}
        if not staged:
            out = self.run(ro, rd, dn, **kwargs)
        else:
            mb = int(max_ray_batch or self.max_ray_batch_default)
            parts = [self.run(ro[a:a + mb], rd[a:a + mb], dn[a:a + mb], **kwargs) for a in range(0, ro.shape[0], mb)]
            out = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
        res = {}
        for k, v in out.items():
            res[k] = v.reshape(*prefix, *v.shape[1:]) if v.dim() > 1 else v.reshape(*prefix)
        return res

    # --- device binding of the flat parameter buffer (see models.ALNetwork)
    def _ensure_device(self):
        raise NotImplementedError
