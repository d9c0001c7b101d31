"""Pure-PyTorch fp32 restatement of the NeRF half of autolabel's hot path (plain torch ops, CPU by default; the tensors may
live on any torch device -- the matched-quality gate runs it on the GPU box's device through torch's own kernels, never through
this repo's HIP library).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **Parity unpinned**: the
arithmetic restated here lives in two dependencies that are absent from
/root/reference (tinycudann, unpinned; ethz-asl/torch-ngp fork, unpinned empty
submodule).  What IS anchored on the reference:

* wiring of encoders / heads        autolabel/models.py:30-59, 64-136
* density / color / semantic        autolabel/models.py:175-220, 248-256
* instantiated sizes                autolabel/model_utils.py:61-74
* 4-term loss                       autolabel/trainer.py:54-94
* optimizer                         scripts/train.py:50-63

External algorithms restated (published behaviour, frozen here as the spec):

* tiny-cuda-nn ``GridEncoding`` (hash, Linear interpolation), ``Frequency``,
  ``SphericalHarmonics`` deg 4, bias-free ReLU MLPs whose input is padded to a
  multiple of 16 with ONES and whose output is padded to a multiple of 16.
* torch-ngp ``NeRFRenderer.run`` (non-cuda-ray branch), ``sample_pdf``,
  ``near_far_from_aabb``, ``trunc_exp``.
* fork additions (``direction_norms``, ``semantic``, ``semantic_features``,
  ``depth_variance``, ``coordinates_map``) are DEFINED here, see DESIGN.md.

Spec decisions that differ from a literal reading of upstream are marked
``SPEC:``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

PI = math.pi
PAD_VALUE = 1.0  # tcnn pads network inputs with ones (acts as a bias column)


# --------------------------------------------------------------------------- RNG
# Counter-based generator shared bit-for-bit with the HIP kernels
# (autolabel_amd/csrc/common.h: aln_rand_u32).  uint32 wrap-around arithmetic.
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def rand_u32(seed: int, stream: int, step: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        key = _fmix32(np.array([(seed + 0x9E3779B9 * (stream + 1)) & 0xFFFFFFFF], dtype=np.uint32))
        key = _fmix32(key ^ np.uint32(step & 0xFFFFFFFF))
        idx = np.asarray(idx).astype(np.uint32)
        r = _fmix32(idx * np.uint32(0x9E3779B1) + key)
        r = _fmix32(r ^ np.uint32(0x68E31DA4))
    return r


def rand_uniform(seed: int, stream: int, step: int, idx: np.ndarray) -> np.ndarray:
    """Uniform [0,1) float32 with 24 random bits."""
    return (rand_u32(seed, stream, step, idx) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


STREAM_FRAME, STREAM_PIXEL, STREAM_JX, STREAM_JY, STREAM_PERTURB, STREAM_PDF, STREAM_CLASS = range(7)


# ------------------------------------------------------------------- hash grid
@dataclass
class GridSpec:
    """tcnn GridEncoding configuration at autolabel/models.py:38-48."""
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap_size: int = 19
    base_resolution: int = 16
    per_level_scale: float = 2.0
    pos_fma: bool = False    # True: pos = fma(x, scale, 0.5) (one rounding, what tcnn's kernel compiles to) instead of the unfused SPEC

    def levels(self):
        out, offset = [], 0
        for l in range(self.n_levels):
            # tcnn grid.h: scale = exp2(l*log2(pls))*base - 1 ; res = ceil(scale)+1   (fp32)
            scale = np.float32(np.exp2(np.float32(l) * np.log2(np.float32(self.per_level_scale))) *
                               np.float32(self.base_resolution) - np.float32(1.0))
            res = int(np.ceil(scale)) + 1
            dense_size = res ** 3
            size = min(dense_size, 0x7FFFFFFF)
            size = (size + 7) // 8 * 8
            size = min(size, 1 << self.log2_hashmap_size)
            # dense indexing is used while the running stride stays <= size
            dense = (res * res * res) <= size
            out.append(dict(scale=float(scale), res=res, size=size, offset=offset, dense=dense, pos_fma=self.pos_fma))
            offset += size
        return out

    @property
    def n_entries(self):
        lv = self.levels()
        return lv[-1]['offset'] + lv[-1]['size']

    @property
    def out_dim(self):
        return self.n_levels * self.n_features


PRIME_Y = 2654435761
PRIME_Z = 805459861


_CORNER_BITS = [[(c >> d) & 1 for d in range(3)] for c in range(8)]   # corner c uses +1 along dim d iff bit d of c is set


def grid_corner_indices(xn: torch.Tensor, level: dict):
    """xn [M,3] fp32 in [0,1].  Returns (idx [M,8] int64 level-local, w [M,8] fp32).

    tcnn ``pos_fract`` / ``grid_index``.  SPEC: pos = x*scale + 0.5 is evaluated
    UNFUSED in fp32 (two roundings) so that floor() is reproducible bit-exactly.
    Corner c uses +1 along dim d iff bit d of c is set; accumulation order c=0..7.
    (All eight corners are evaluated as one [M,8] batch: element for element the arithmetic
    of the per-corner loop -- w = ((1*w_x)*w_y)*w_z, uint32 hash -- just fewer tensor ops.)
    """
    dev = xn.device
    scale = torch.tensor(level['scale'], dtype=torch.float32, device=dev)
    if level.get('pos_fma'):   # x * scale (48 significant bits) + 0.5 is exact in fp64: one rounding to fp32 = the fused result
        pos = (xn.double() * scale.double() + 0.5).float()
    else:
        pos = xn * scale + 0.5
    g = torch.floor(pos)
    frac = pos - g
    g = g.to(torch.int64)
    res, size = level['res'], level['size']
    bits = torch.tensor(_CORNER_BITS, dtype=torch.int64, device=dev)               # [8,3]
    cg = g[:, None, :] + bits[None]                                                # [M,8,3]
    wd = torch.where(bits[None].bool(), frac[:, None, :], 1.0 - frac[:, None, :])  # [M,8,3]
    w = torch.ones(xn.shape[0], 8, dtype=torch.float32, device=dev)
    for d in range(3):
        w = w * wd[..., d]
    if level['dense']:
        idx = cg[..., 0] + cg[..., 1] * res + cg[..., 2] * res * res
    else:
        # uint32 arithmetic
        idx = ((cg[..., 0] & 0xFFFFFFFF) ^ ((cg[..., 1] * PRIME_Y) & 0xFFFFFFFF) ^ ((cg[..., 2] * PRIME_Z) & 0xFFFFFFFF))
    return idx % size, w


class _RoundHalf(torch.autograd.Function):
    """Round to fp16 in forward, straight-through fp32 gradient."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x, half_sim):
    return _RoundHalf.apply(x) if half_sim else x


def hashgrid_encode(xn: torch.Tensor, table: torch.Tensor, spec: GridSpec, half_sim=False):
    """xn [M,3] in [0,1]; table [n_entries, F].  Output [M, L*F], level-major."""
    tab = _q(table, half_sim)
    outs = []
    for level in spec.levels():
        idx, w = grid_corner_indices(xn, level)
        prod = w[..., None] * tab[level['offset'] + idx]        # [M,8,F]: one gather for the eight corners
        acc = prod[:, 0]                                         # (0 + x = x: the running sum starts at corner 0)
        for c in range(1, 8):                                    # accumulation order c = 0..7
            acc = acc + prod[:, c]
        outs.append(acc)
    return torch.cat(outs, 1)


def freq_encode(x: torch.Tensor, n_freq: int):
    """tcnn Frequency: out[d*2n + 2k + {0,1}] = sin(2^k*pi*x_d + {0, pi/2})."""
    outs = []
    for d in range(x.shape[1]):
        for k in range(n_freq):
            arg = x[:, d] * np.float32(2.0 ** k) * np.float32(PI)
            outs.append(torch.sin(arg))
            outs.append(torch.sin(arg + np.float32(PI / 2)))
    return torch.stack(outs, 1)


def sh4_encode(d01: torch.Tensor):
    """tcnn SphericalHarmonics degree 4 on inputs in [0,1] (remapped to [-1,1])."""
    v = d01 * 2.0 - 1.0
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(o, 1)


class _TruncExp(torch.autograd.Function):
    """torch-ngp activation.trunc_exp: fwd exp(x), bwd g*exp(clamp(x,-15,15))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def pad16(n):
    return (n + 15) // 16 * 16


def mlp_forward(x: torch.Tensor, weights: List[torch.Tensor], half_sim=False):
    """Bias-free ReLU MLP; weights[i] is [out_pad_i, in_pad_i] (y = W x).

    Input padded with ONES to in_pad_0; returns the full padded output.
    With half_sim, weights and every layer output are rounded to fp16
    (fp32 accumulate), mirroring tcnn FullyFusedMLP.
    """
    in_pad = weights[0].shape[1]
    if x.shape[1] < in_pad:
        x = torch.cat([x, torch.full((x.shape[0], in_pad - x.shape[1]), PAD_VALUE, dtype=x.dtype, device=x.device)], 1)
    h = _q(x, half_sim)
    for i, W in enumerate(weights):
        h = h @ _q(W, half_sim).t()
        if i + 1 < len(weights):
            h = torch.relu(h)
        h = _q(h, half_sim)
    return h


# ------------------------------------------------------------------ the model
@dataclass
class ModelConfig:
    encoding: str = 'hg+freq'
    geo_feat_dim: int = 15
    hidden_dim: int = 128
    hidden_dim_color: int = 128
    feature_dim: int = 64  # hidden_dim_semantic
    n_classes: int = 2
    bound: float = 1.0
    density_scale: float = 1.0
    min_near: float = 0.2
    grid: GridSpec = field(default_factory=GridSpec)

    @property
    def n_freq(self):
        return {'hg+freq': 2, 'freq': 10, 'hg': 0}[self.encoding]

    @property
    def enc_dim(self):
        g = self.grid.out_dim if self.encoding != 'freq' else 0
        return 3 * 2 * self.n_freq + g


def mlp_shapes(cfg: ModelConfig) -> Dict[str, List[tuple]]:
    """Padded [out,in] weight shapes per head (autolabel/model_utils.py:61-74)."""
    G, D, C = cfg.geo_feat_dim, cfg.feature_dim, cfg.n_classes
    H, Hc = cfg.hidden_dim, cfg.hidden_dim_color
    return {
        'sigma': [(H, pad16(cfg.enc_dim)), (H, H), (pad16(1 + G), H)],
        'color': [(Hc, pad16(16 + G)), (Hc, Hc), (pad16(3), Hc)],
        'semf': [(D, pad16(G)), (D, D), (pad16(D), D)],
        'semo': [(64, pad16(D + G)), (pad16(C), 64)],
    }


def init_params(cfg: ModelConfig, seed=0) -> Dict[str, torch.Tensor]:
    """tcnn defaults: grid U(-1e-4,1e-4); MLP xavier-uniform."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    if cfg.encoding != 'freq':
        p['grid'] = (torch.rand(cfg.grid.n_entries, cfg.grid.n_features, generator=g) * 2 - 1) * 1e-4
    for name, shapes in mlp_shapes(cfg).items():
        for i, (o, n) in enumerate(shapes):
            lim = math.sqrt(6.0 / (o + n))
            p[f'{name}.{i}'] = (torch.rand(o, n, generator=g) * 2 - 1) * lim
    return p


class OracleModel:
    """Functional restatement of ALNetwork (autolabel/models.py:62-265)."""

    def __init__(self, cfg: ModelConfig, params: Optional[Dict[str, torch.Tensor]] = None, half_sim=False, seed=0, device=None):
        self.cfg = cfg
        self.half_sim = half_sim
        self.params = params if params is not None else init_params(cfg, seed)
        if device is not None:
            self.params = {k: v.detach().to(device) for k, v in self.params.items()}
        for v in self.params.values():
            v.requires_grad_(True)

    def _w(self, name):
        return [self.params[f'{name}.{i}'] for i in range(len(mlp_shapes(self.cfg)[name]))]

    # models.py:25-27, 51-59
    def encode(self, x):
        cfg = self.cfg
        parts = []
        if cfg.encoding == 'freq':
            parts.append(freq_encode((x + cfg.bound) / (2.0 * cfg.bound), 10))
        else:
            if cfg.encoding == 'hg+freq':
                parts.append(freq_encode(x, 2))  # raw, un-normalised x (models.py:52)
            xn = torch.clip((x + cfg.bound) / (2.0 * cfg.bound), 0.0, 1.0)
            parts.append(hashgrid_encode(xn, self.params['grid'], cfg.grid, self.half_sim))
        return torch.cat(parts, 1)

    # models.py:175-188
    def density(self, x):
        h = mlp_forward(self.encode(x), self._w('sigma'), self.half_sim)
        G = self.cfg.geo_feat_dim
        return {'sigma': trunc_exp(h[:, 0]), 'geo_feat': h[:, 1:1 + G]}

    # models.py:190-220
    def color(self, x, d, mask=None, geo_feat=None):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=torch.float32, device=x.device)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        enc_d = sh4_encode((d + 1) / 2)
        h = mlp_forward(torch.cat([enc_d, geo_feat], 1), self._w('color'), self.half_sim)[:, :3]
        h = torch.sigmoid(h)
        if mask is not None:
            rgbs = rgbs.index_put((mask.nonzero(as_tuple=True)[0],), h)
            return rgbs
        return h

    # models.py:248-256
    def semantic(self, geo_feat, sigma=None):
        D, C = self.cfg.feature_dim, self.cfg.n_classes
        f = mlp_forward(geo_feat, self._w('semf'), self.half_sim)[:, :D]
        logits = mlp_forward(torch.cat([torch.relu(f), geo_feat], 1), self._w('semo'), self.half_sim)[:, :C]
        return logits, f

    # ------------------------------------------------------------ renderer
    def near_far(self, rays_o, rays_d):
        """torch-ngp raymarching.near_far_from_aabb, aabb = [-bound,bound]^3.

        SPEC: per-axis slab via fmin/fmax (NaN-ignoring); a miss gives
        near = far = min_near (upstream: FLT_MAX); far = max(far, near).
        """
        b = self.cfg.bound
        rd = 1.0 / rays_d
        t1 = (-b - rays_o) * rd
        t2 = (b - rays_o) * rd
        tn = torch.fmin(t1, t2)
        tf = torch.fmax(t1, t2)
        near = torch.fmax(torch.fmax(tn[:, 0], tn[:, 1]), tn[:, 2])
        far = torch.fmin(torch.fmin(tf[:, 0], tf[:, 1]), tf[:, 2])
        miss = ~(near <= far)
        mn = torch.tensor(self.cfg.min_near, dtype=torch.float32, device=rays_o.device)
        near = torch.where(miss, mn, torch.fmax(near, mn))
        far = torch.where(miss, mn, far)
        far = torch.fmax(far, near)
        return near, far

    def _weights(self, z, sigma, sample_dist):
        deltas = torch.cat([z[:, 1:] - z[:, :-1], sample_dist], 1)
        alphas = 1 - torch.exp(-deltas * self.cfg.density_scale * sigma)
        shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-15], 1)
        T = torch.cumprod(shifted, 1)[:, :-1]
        return alphas * T, alphas, T, deltas

    def run(self, rays_o, rays_d, direction_norms, num_steps=128, upsample_steps=128, bg_color=None,
            perturb=False, noise_coarse=None, u_fine=None, want_semantic=True, z_fine_override=None):
        """torch-ngp NeRFRenderer.run (non-cuda-ray) + fork outputs.

        noise_coarse [N,num_steps] / u_fine [N,upsample_steps]: explicit uniform
        randoms (used when perturb=True); the HIP kernels consume the same
        numbers so parity tests are deterministic.  z_fine_override [N,upsample_steps]
        replaces the importance samples (finest grid cells are ~4e-6 wide, so gradient parity per
        table entry needs bit-identical sample positions; the sampler itself is tested separately).
        """
        cfg = self.cfg
        N = rays_o.shape[0]
        near, far = self.near_far(rays_o, rays_d)
        near, far = near[:, None], far[:, None]
        lin = torch.arange(num_steps, dtype=torch.float32, device=rays_o.device) / np.float32(max(num_steps - 1, 1))
        z = near + (far - near) * lin[None]
        sample_dist = (far - near) / np.float32(num_steps)
        if perturb:
            z = z + (noise_coarse - 0.5) * sample_dist

        def pts(zv):
            p = rays_o[:, None, :] + rays_d[:, None, :] * zv[..., None]
            return torch.clamp(p, -cfg.bound, cfg.bound)

        xyz = pts(z)
        dens = self.density(xyz.reshape(-1, 3))
        sigma = dens['sigma'].view(N, num_steps)
        geo = dens['geo_feat'].view(N, num_steps, -1)

        if upsample_steps > 0:
            with torch.no_grad():
                w, _, _, deltas = self._weights(z, sigma, sample_dist)
                z_mid = z[:, :-1] + 0.5 * deltas[:, :-1]
                if perturb:
                    u = u_fine
                else:
                    u = (torch.arange(upsample_steps, dtype=torch.float32, device=rays_o.device) + 0.5) / np.float32(upsample_steps)
                    u = u[None].expand(N, upsample_steps)
                new_z = sample_pdf(z_mid, w[:, 1:-1], u.contiguous())
                if z_fine_override is not None:
                    new_z = z_fine_override
                new_xyz = pts(new_z)
            nd = self.density(new_xyz.reshape(-1, 3))
            z = torch.cat([z, new_z], 1)
            z, order = torch.sort(z, dim=1, stable=True)
            xyz = torch.gather(torch.cat([xyz, new_xyz], 1), 1, order[..., None].expand(-1, -1, 3))
            sigma = torch.gather(torch.cat([sigma, nd['sigma'].view(N, upsample_steps)], 1), 1, order)
            geo = torch.cat([geo, nd['geo_feat'].view(N, upsample_steps, -1)], 1)
            geo = torch.gather(geo, 1, order[..., None].expand(-1, -1, geo.shape[-1]))

        S = z.shape[1]
        weights, alphas, T, deltas = self._weights(z, sigma, sample_dist)
        mask = weights > 1e-4
        dirs = rays_d[:, None, :].expand(N, S, 3)
        rgbs = self.color(xyz.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1),
                          geo_feat=geo.reshape(N * S, -1)).view(N, S, 3)
        wsum = weights.sum(1)
        zd = z / direction_norms.view(N, 1)  # SPEC (fork): metric z-depth
        depth = (weights * zd).sum(1)
        depth_var = (weights * (zd - depth[:, None]) ** 2).sum(1)
        image = (weights[..., None] * rgbs).sum(1)
        bg = 1.0 if bg_color is None else bg_color  # upstream: None -> 1
        image = image + (1 - wsum)[:, None] * bg
        out = {'image': image, 'depth': depth, 'weights_sum': wsum, 'depth_variance': depth_var,
               'coordinates_map': (weights[..., None] * xyz).sum(1),
               '_weights': weights, '_z': z, '_mask': mask, '_sigma': sigma}
        if want_semantic:
            # SPEC (fork): heads evaluated on every sample, composited with the same weights
            logits, f = self.semantic(geo.reshape(N * S, -1))
            out['semantic'] = (weights[..., None] * logits.view(N, S, -1)).sum(1)
            out['semantic_features'] = (weights[..., None] * f.view(N, S, -1)).sum(1)
        return out


def sample_pdf(bins, weights, u):
    """torch-ngp nerf/renderer.py sample_pdf with explicit u."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


# ------------------------------------------------------------------------ loss
DEPTH_EPSILON = 0.01


def loss_fn(out, batch, rgb_weight=1.0, depth_weight=0.1, semantic_weight=1.0, feature_weight=0.5,
            feature_loss=False):
    """autolabel/trainer.py:54-94.  SPEC: empty depth/label sets contribute 0
    (torch's mean over an empty set would be NaN)."""
    terms = {}
    terms['rgb'] = ((out['image'] - batch['pixels']) ** 2).mean()
    loss = rgb_weight * terms['rgb']
    has_depth = batch['depth'] > DEPTH_EPSILON
    if has_depth.any():
        terms['depth'] = (out['depth'][has_depth] - batch['depth'][has_depth]).abs().mean()
        loss = loss + depth_weight * terms['depth']
    if feature_loss:
        gt = batch['features']
        terms['feature'] = (out['semantic_features'][:, :gt.shape[1]] - gt).abs().mean()
        loss = loss + feature_weight * terms['feature']
    has_sem = batch['semantic'] >= 0
    if has_sem.any():
        terms['semantic'] = torch.nn.functional.cross_entropy(out['semantic'][has_sem], batch['semantic'][has_sem])
        loss = loss + semantic_weight * terms['semantic']
    return loss, terms


def adam_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.99, eps=1e-15, weight_decay=0.0):
    """torch.optim.Adam single-tensor step (scripts/train.py:50-63). In place."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
