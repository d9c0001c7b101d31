"""``ALNetwork`` and its encoders on the MI355X HIP path (interface of autolabel/models.py).

The reference builds every sub-network from tinycudann modules (autolabel/models.py:19,34,38,84,97,104,117,127); here each
sub-module is a thin ``nn.Module`` that owns one flat fp32 ``params`` tensor (the tcnn convention, so ``state_dict`` keys
look the same: ``sigma_net.params``, ``encoder.grid_encoding.params`` ...).  Once the model is on a GPU the blocks become
views into ONE flat master buffer (``pipeline.Params``) next to an fp16 table shadow and MFMA-fragment copies of the MLP
weights; all arithmetic runs in the HIP kernels.  There is no CPU compute path: calling ``density``/``color``/``semantic``/
``render`` on a CPU model raises.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import hip as H
from .pipeline import HipPipeline, ModelLayout, Params, f16
from .renderer import NeRFRenderer


class _Block(nn.Module):
    """Parameter block with tcnn-like attributes."""

    def __init__(self, n_params, n_input_dims, n_output_dims):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(n_params, dtype=torch.float32))
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims


class FreqEncoder(nn.Module):
    """10-frequency encoding of (x+b)/2b (autolabel/models.py:15-27)."""

    def __init__(self, input_dim):
        super().__init__()
        self.encoder = _Block(0, input_dim, input_dim * 2 * 10)
        self.n_output_dims = self.encoder.n_output_dims


class HGFreqEncoder(nn.Module):
    """cat[Frequency(n=2)(x), HashGrid(clip((x+b)/2b))] (autolabel/models.py:30-59)."""

    def __init__(self, input_dim, grid=None):
        super().__init__()
        g = grid if grid is not None else H.make_grid_desc()
        self.encoder = _Block(0, input_dim, input_dim * 2 * 2)
        self.grid_encoding = _Block(int(g.n_entries) * g.n_features, input_dim, g.n_levels * g.n_features)
        self.n_output_dims = self.encoder.n_output_dims + self.grid_encoding.n_output_dims


class HashGridEncoder(nn.Module):
    """``get_encoder('hashgrid', desired_resolution=2**18)`` of torch-ngp (autolabel/models.py:142-143)."""

    def __init__(self, input_dim, grid):
        super().__init__()
        self.grid_encoding = _Block(int(grid.n_entries) * grid.n_features, input_dim, grid.n_levels * grid.n_features)
        self.n_output_dims = self.grid_encoding.n_output_dims


class ALNetwork(NeRFRenderer):

    def __init__(self, encoding='hg', num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64,
                 hidden_dim_semantic=64, semantic_classes=2, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.hidden_dim_semantic, self.semantic_classes = hidden_dim_semantic, semantic_classes
        self.encoding = encoding
        grid = None
        self.tcnn_fma = bool(kwargs.pop('tcnn_fma', False)) if 'tcnn_fma' in kwargs else False
        if encoding == 'hg':
            # torch-ngp hashgrid: 16 levels, base 16 up to desired_resolution 2**18 * bound
            pls = float(2.0 ** (math.log2(2 ** 18 * bound / 16) / 15))
            grid = H.make_grid_desc(per_level_scale=pls)
        self._layout = ModelLayout(encoding, geo_feat_dim, hidden_dim, hidden_dim_color, hidden_dim_semantic, semantic_classes,
                                   num_layers=num_layers, num_layers_color=num_layers_color, bound=float(bound), grid=grid)
        self.set_tcnn_fma(self.tcnn_fma)
        L = self._layout
        self.encoder, self.in_dim = self._get_encoder(encoding)
        nets = L.nets
        self.sigma_net = _Block(nets['sigma'].n_params, self.in_dim, 1 + geo_feat_dim)
        self.encoder_dir = _Block(0, 3, 16)
        self.color_features = 16 + geo_feat_dim
        self.color_net = _Block(nets['color'].n_params, self.color_features, 3)
        self.semantic_features = _Block(nets['semf'].n_params, geo_feat_dim, hidden_dim_semantic)
        self.semantic_out = _Block(nets['semo'].n_params, hidden_dim_semantic + geo_feat_dim, semantic_classes)
        self._P = self._pipe = None
        self._seed = 0
        self._shadow_version = None
        self.reset_parameters()

    def set_tcnn_fma(self, on=True):
        """Grid position as ONE fused multiply-add (what tcnn's kernel compiles `x * scale + 0.5` to) instead of this build's
        two-rounding spec.  Fields trained by the reference were fitted through the fused form: model_utils.load_checkpoint
        (reference=True) switches it on so their finest-level cells are hit exactly as during training."""
        self.tcnn_fma = bool(on)
        self._layout.enc.grid.pos_fma = int(self.tcnn_fma)

    def _get_encoder(self, encoding):
        if encoding == 'freq':
            enc = FreqEncoder(3)
        elif encoding == 'hg':
            enc = HashGridEncoder(3, self._layout.enc.grid)
        elif encoding == 'hg+freq':
            enc = HGFreqEncoder(3)
        else:
            raise NotImplementedError(f'Unknown input encoding {encoding}')
        return enc, enc.n_output_dims

    # ------------------------------------------------------------------ parameters
    def _param_blocks(self):
        blocks = []
        if self._layout.n_grid:
            blocks.append(('grid', self.encoder.grid_encoding.params))
        blocks += [('sigma', self.sigma_net.params), ('color', self.color_net.params),
                   ('semf', self.semantic_features.params), ('semo', self.semantic_out.params)]
        return blocks

    def _block_range(self, name):
        L = self._layout
        if name == 'grid':
            return 0, L.n_grid
        return L.offsets[name], L.offsets[name] + L.nets[name].n_params

    def reset_parameters(self, seed=0):
        """tcnn defaults: grid U(-1e-4, 1e-4); MLP matrices xavier-uniform on their padded [out, in] shape."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self._param_blocks():
                if name == 'grid':
                    p.copy_(((torch.rand(p.numel(), generator=g) * 2 - 1) * 1e-4).to(p.device))
                else:
                    o = 0
                    for (no, ni) in self._layout.nets[name].shapes:
                        lim = math.sqrt(6.0 / (no + ni))
                        p[o:o + no * ni] = ((torch.rand(no * ni, generator=g) * 2 - 1) * lim).to(p.device)
                        o += no * ni

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._P = self._pipe = None  # storage moved: re-bind lazily
        return out

    def _ensure_device(self):
        """Bind the parameter blocks to one flat device buffer and keep the fp16 shadows current."""
        p0 = self.sigma_net.params
        if not p0.is_cuda:
            H.require_gpu()
            raise RuntimeError('ALNetwork is on the CPU: autolabel_amd has no CPU compute path, call .cuda() first')
        H.lib()
        if self._P is None:
            P = Params(self._layout, p0.device)
            with torch.no_grad():
                for name, p in self._param_blocks():
                    a, b = self._block_range(name)
                    P.flat[a:b].copy_(p.data.reshape(-1))
                    p.data = P.flat[a:b]
            self._P, self._pipe = P, HipPipeline(self._layout, P, density_scale=float(self.density_scale), min_near=float(self.min_near))
            if self.cuda_ray:
                from .pipeline import OccupancyGrid
                self._pipe.occ = OccupancyGrid(p0.device, G=self.grid_size, max_steps=self.max_steps, samples=self.march_samples,
                                               density_thresh=float(self.density_thresh), grid=self.density_grid,
                                               bits=self.density_bitfield)
                self._pipe.recount_bitfield()    # a loaded checkpoint carries grid AND bitfield: the bits are kept as stored
            self._shadow_version = None
        ver = tuple(p._version for _, p in self._param_blocks())
        if ver != self._shadow_version:
            self._P.refresh_shadows()
            self._shadow_version = ver
        return self._pipe

    def state_dict(self, *args, **kwargs):
        if self._P is not None and self._P.masters_stale:
            raise RuntimeError('ALNetwork.state_dict: the hash table\'s fp32 masters are sharded over the ranks (TrainEngine(shard_optimizer=True)); '
                               'call TrainEngine.sync_master() on every rank first (Trainer.save_checkpoint does)')
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        res = super().load_state_dict(state_dict, strict=strict)
        self._shadow_version = None
        return res

    def get_params(self, lr):
        blocks = [self.encoder, self.sigma_net, self.encoder_dir, self.color_net, self.semantic_features, self.semantic_out]
        return [{'params': b.parameters(), 'lr': lr} for b in blocks]

    def network_parameters(self):
        """MLP parameters, excluding the encoder (the weight-decayed group of scripts/train.py:55-58)."""
        return (list(self.sigma_net.parameters()) + list(self.color_net.parameters()) +
                list(self.semantic_features.parameters()) + list(self.semantic_out.parameters()))

    # ------------------------------------------------------------------ point queries (inference)
    def _sigma_rows(self, x):
        pipe, L = self._ensure_device(), self._layout
        x = x.reshape(-1, 3).float().contiguous()
        n = x.shape[0]
        enc = torch.empty(n, L.enc.enc_pad, dtype=f16, device=x.device)
        out = torch.empty(n, 16, dtype=f16, device=x.device)
        sigma = torch.empty(n, dtype=torch.float32, device=x.device)
        pipe.density_rows(n, None, None, None, x, 1, enc, None, None, out, sigma, train=False)
        return out, sigma

    @torch.no_grad()
    def density(self, x):
        """x: [N,3] in [-bound, bound] -> {'sigma': [N], 'geo_feat': [N, G]}   (autolabel/models.py:175-188)."""
        out, sigma = self._sigma_rows(x)
        return {'sigma': sigma, 'geo_feat': out[:, 1:1 + self.geo_feat_dim]}

    @torch.no_grad()
    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """rgb in [0,1] for the rows selected by mask, zeros elsewhere (autolabel/models.py:190-220)."""
        pipe, L = self._ensure_device(), self._layout
        n = x.shape[0]
        rgbs = torch.zeros(n, 3, dtype=torch.float32, device=x.device)
        sel = torch.arange(n, device=x.device) if mask is None else mask.nonzero(as_tuple=True)[0]
        if sel.numel() == 0:
            return rgbs
        m = sel.numel()
        so = torch.zeros(m, 16, dtype=f16, device=x.device)
        so[:, 1:1 + L.G] = geo_feat[sel].to(f16)
        dirs = d[sel].float().contiguous()
        cs = L.nets['color']
        cin = torch.empty(m, cs.in_pad, dtype=f16, device=x.device)
        cout = torch.empty(m, cs.out_pad, dtype=f16, device=x.device)
        H.call('aln_build_color_in', None, None, m, None, H.ptr(dirs), 0, 1, 1, H.ptr(so), L.G, cs.in_pad, H.ptr(cin), H.stream())
        H.call('aln_mlp_fwd', C.byref(pipe.P.descs['color']), H.ptr(cin), m, None, None, None, H.ptr(cout), H.stream())
        rgbs[sel] = torch.sigmoid(cout[:, :3].float())
        return rgbs

    @torch.no_grad()
    def semantic(self, geo_features, sigma=None):
        """-> (logits [N,C], semantic features [N,D])   (autolabel/models.py:248-256)."""
        pipe, L = self._ensure_device(), self._layout
        n = geo_features.shape[0]
        dev = geo_features.device
        so = torch.zeros(n, 16, dtype=f16, device=dev)
        so[:, 1:1 + L.G] = geo_features.to(f16)
        fs, os_ = L.nets['semf'], L.nets['semo']
        if L.sem_wide:     # LSeg-width heads: wide.hip GEMMs, inputs built inside the kernels
            logits, feat, _ = pipe.wide_sem_fwd(so, n, lambda name, shp: torch.empty(shp, dtype=f16, device=dev))
            return logits[:, :L.C], feat[:, :L.D]
        fin = torch.empty(n, fs.in_pad, dtype=f16, device=dev)
        feat = torch.empty(n, fs.out_pad, dtype=f16, device=dev)
        oin = torch.empty(n, os_.in_pad, dtype=f16, device=dev)
        logits = torch.empty(n, os_.out_pad, dtype=f16, device=dev)
        H.call('aln_build_sem_in', H.ptr(so), None, n, L.D, L.G, fs.in_pad, os_.in_pad, H.ptr(fin), None, H.stream())
        H.call('aln_mlp_fwd', C.byref(pipe.P.descs['semf']), H.ptr(fin), n, None, None, None, H.ptr(feat), H.stream())
        H.call('aln_build_sem_in', H.ptr(so), H.ptr(feat), n, L.D, L.G, fs.in_pad, os_.in_pad, None, H.ptr(oin), H.stream())
        H.call('aln_mlp_fwd', C.byref(pipe.P.descs['semo']), H.ptr(oin), n, None, None, None, H.ptr(logits), H.stream())
        return logits[:, :L.C], feat[:, :L.D]

    @torch.no_grad()
    def forward(self, x, d):
        """(sigma, rgb, softmax(semantic)) per point; geo_feat goes through ReLU here (autolabel/models.py:150-173)."""
        out, sigma = self._sigma_rows(x)
        geo = torch.relu(out[:, 1:1 + self.geo_feat_dim])
        # forward() hands d straight to the SH encoding (no (d+1)/2 remap, models.py:161), i.e. SH sees 2d-1
        rgb = self.color(x, 2 * d.float() - 1, geo_feat=geo)
        logits, _ = self.semantic(geo)
        return sigma, rgb, torch.softmax(logits.float(), dim=-1)


class Autoencoder(nn.Module):
    """Feature-compression autoencoder used offline by scripts/compute_feature_maps.py (autolabel/models.py:268-294).
    Out of the hot path: plain torch modules with the reference's layer sizes."""

    def __init__(self, in_features, bottleneck):
        super().__init__()
        self.encoder = nn.Sequential(nn.Linear(in_features, 128, bias=False), nn.ReLU(), nn.Linear(128, bottleneck, bias=False), nn.ReLU())
        self.decoder = nn.Sequential(nn.Linear(bottleneck, 128, bias=False), nn.ReLU(), nn.Linear(128, in_features, bias=False))

    def forward(self, x, p=0.1):
        code = self.encoder(x)
        return self.decoder(torch.nn.functional.dropout(code, 0.1)), code
