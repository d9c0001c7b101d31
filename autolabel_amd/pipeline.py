"""Host-side orchestration of the HIP kernels for one render / training pass.

This is the glue the reference gets from torch-ngp's ``NeRFRenderer.run`` (a chain of PyTorch ops) plus
tinycudann's module wrappers; here it is a fixed sequence of C-ABI launches on the current HIP stream
with caller-owned workspaces (no allocation inside the step once shapes are warm, no host sync).

Parameter memory: one flat fp32 master buffer ``[hash-grid table | sigma | color | semf | semo]`` with a
matching flat gradient buffer (a single RCCL all-reduce covers every gradient), an fp16 shadow of the
table (what the gather kernel reads) and fp16 MFMA-fragment copies of the MLP weights.
"""
import ctypes as C
import math

import torch

from . import hip as H


def pad16(n):
    return (n + 15) // 16 * 16


class MlpSpec:
    """Padded shape of one bias-free ReLU MLP (tcnn Network; autolabel/models.py:84-136)."""

    def __init__(self, name, n_in, hidden, n_out, n_hidden):
        self.name, self.n_in, self.hidden, self.n_out, self.n_hidden = name, n_in, hidden, n_out, n_hidden
        self.in_pad, self.out_pad = pad16(n_in), pad16(n_out)
        self.shapes = [(hidden, self.in_pad)] + [(hidden, hidden)] * (n_hidden - 1) + [(self.out_pad, hidden)]
        self.n_params = sum(o * i for o, i in self.shapes)
        if n_hidden not in (1, 2):
            raise NotImplementedError(
                f'{name}: {n_hidden} hidden layers are not supported by the HIP heads (1-2).  Note: the reference class default '
                'num_layers_color=3 (autolabel/models.py:70) is never used by the reference itself -- create_model passes 2 '
                '(autolabel/model_utils.py:66); construct ALNetwork with num_layers_color=2.')
        # Heads too wide for the register-chained MFMA kernels of mlp.hip (weights resident in LDS) -- LSeg: D=512 -> semf
        # 16->512->512->512, semo 528->64->C; or hundreds of classes -- run layer by layer on the hand-written MFMA GEMMs of
        # wide.hip (weights streamed from L2, inputs / ReLU / masks / accumulation fused into prologue and epilogue).
        self.wide = hidden not in (64, 128) or self.in_pad > 96 or self.out_pad > 64


class ModelLayout:
    """Flat parameter layout for ALNetwork (sizes from autolabel/model_utils.py:61-74)."""

    def __init__(self, encoding, geo_feat_dim, hidden_dim, hidden_dim_color, hidden_dim_semantic, semantic_classes,
                 num_layers=2, num_layers_color=2, bound=1.0, grid=None):
        self.enc = H.make_enc_desc(encoding, float(bound), grid)
        self.G, self.D, self.C = geo_feat_dim, hidden_dim_semantic, semantic_classes
        if not 1 <= geo_feat_dim <= 15:
            raise NotImplementedError('geo_feat_dim must be in 1..15 for the HIP heads (density + geo_feat share one 16-wide row)')
        self.n_grid = int(self.enc.grid.n_entries) * 2 if self.enc.use_grid else 0
        self.nets = {
            'sigma': MlpSpec('sigma', self.enc.enc_dim, hidden_dim, 1 + geo_feat_dim, num_layers),
            'color': MlpSpec('color', 16 + geo_feat_dim, hidden_dim_color, 3, num_layers_color),
            'semf': MlpSpec('semf', geo_feat_dim, hidden_dim_semantic, hidden_dim_semantic, 2),
            'semo': MlpSpec('semo', hidden_dim_semantic + geo_feat_dim, 64, semantic_classes, 1),
        }
        self.offsets, o = {}, self.n_grid
        for k, s in self.nets.items():
            self.offsets[k] = o
            o += s.n_params
        self.n_total = o
        self.Cpad = self.nets['semo'].out_pad
        self.sem_wide = self.nets['semf'].wide or self.nets['semo'].wide    # then BOTH semantic heads take the wide path


class Params:
    """Device-resident parameter / gradient / shadow buffers."""

    def __init__(self, layout, device):
        L = self.layout = layout
        self.device = device
        self.flat = torch.zeros(L.n_total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(L.n_total + 8, dtype=torch.float32, device=device)  # +8: side channel for DP (found_inf)
        self.table16 = torch.zeros(max(L.n_grid, 2), dtype=torch.float16, device=device)
        self.frags, self.descs, self.wide_w, self.wide_wt = {}, {}, {}, {}
        self.desc_sigma_tiled = None
        self.desc_sigma_planes = None   # the density head reading PAIR PLANES (AlnMlpDesc.x_tiled = 2: aln_encode_fwd_planes); x_pitch is set per call
        self.wide_wp = None    # semantic_features' second matrix with the columns of every group of 16 in the order the generated first
                               # layer leaves them in (aln_wide_nt_gen, wide.hip: wide_gen_pack)
        # sharded table optimizer (engine.TrainEngine(shard_optimizer=True, shard_gather='table')): the fp32 masters of the slices other
        # ranks own are stale in `flat` until TrainEngine.sync_master() (a collective) -- readers of the masters check this flag
        self.masters_stale = False
        for k, s in L.nets.items():
            if L.sem_wide and k in ('semf', 'semo'):   # row-major fp16 [out, in] per layer (+ transposes for the data gradients)
                self.wide_w[k] = [torch.zeros(o, i, dtype=torch.float16, device=device) for o, i in s.shapes]
                self.wide_wt[k] = [torch.zeros(i, o, dtype=torch.float16, device=device) for o, i in s.shapes]
                if k == 'semf' and s.hidden % 64 == 0 and s.in_pad == 16:
                    self.wide_wp = torch.zeros(s.hidden, s.hidden, dtype=torch.float16, device=device)
                    perm16 = [4 * (q >> 3) + (q & 3) + 8 * ((q & 7) >> 2) for q in range(16)]
                    self._kperm = torch.tensor([16 * (c // 16) + perm16[c % 16] for c in range(s.hidden)], device=device)
                continue
            nf = H.lib().aln_mlp_frag_halves(s.in_pad, s.hidden, s.out_pad, s.n_hidden, 0)
            nb = H.lib().aln_mlp_frag_halves(s.in_pad, s.hidden, s.out_pad, s.n_hidden, 1)
            nr = H.lib().aln_mlp_rowmajor_halves(s.in_pad, s.hidden, s.out_pad, s.n_hidden)
            wf = torch.zeros(nf, dtype=torch.float16, device=device)
            wb = torch.zeros(nb, dtype=torch.float16, device=device)
            wr = torch.zeros((nr + 7) // 8 * 8, dtype=torch.float16, device=device)
            # per-block weight-gradient partial sums of the recompute backward (reduced in a fixed order: no atomics)
            nws = H.lib().aln_mlp_dw_ws_bytes(s.in_pad, s.hidden, s.out_pad, s.n_hidden)
            ws = torch.empty(nws // 4, dtype=torch.float32, device=device)
            self.frags[k] = (wf, wb, wr, ws)
            # defer_dw_reduce: the backward kernels of a step leave their slabs in place, HipPipeline.backward folds all heads' slabs
            # into the gradient buffer with one aln_mlp_dw_reduce_all launch
            self.descs[k] = H.AlnMlpDesc(s.in_pad, s.hidden, s.out_pad, s.n_hidden, wf.data_ptr(), wb.data_ptr(), wr.data_ptr(),
                                         ws.data_ptr(), nws, 1, 0)
            if k == 'sigma' and H.lib().aln_mlp_supports_tiled(s.in_pad, s.hidden, s.out_pad, s.n_hidden):
                # the same head reading its input rows in the tiled layout the level-phased gather writes (AlnMlpDesc.x_tiled)
                self.desc_sigma_tiled = H.AlnMlpDesc(s.in_pad, s.hidden, s.out_pad, s.n_hidden, wf.data_ptr(), wb.data_ptr(), wr.data_ptr(),
                                                     ws.data_ptr(), nws, 1, 1)
                self.desc_sigma_planes = H.AlnMlpDesc(s.in_pad, s.hidden, s.out_pad, s.n_hidden, wf.data_ptr(), wb.data_ptr(), wr.data_ptr(),
                                                      ws.data_ptr(), nws, 1, 2, 0)

    def init_(self, seed=0):
        """tcnn default initialisation: grid U(-1e-4,1e-4), MLP weights xavier-uniform per padded [out,in] matrix."""
        L = self.layout
        g = torch.Generator(device=self.device).manual_seed(seed)
        if L.n_grid:
            self.flat[:L.n_grid] = (torch.rand(L.n_grid, generator=g, device=self.device) * 2 - 1) * 1e-4
        for k, s in L.nets.items():
            o = L.offsets[k]
            for (no, ni) in s.shapes:
                lim = math.sqrt(6.0 / (no + ni))
                self.flat[o:o + no * ni] = (torch.rand(no * ni, generator=g, device=self.device) * 2 - 1) * lim
                o += no * ni
        self.refresh_shadows()

    def net_view(self, k, buf=None):
        buf = self.flat if buf is None else buf
        o = self.layout.offsets[k]
        return buf[o:o + self.layout.nets[k].n_params]

    def grid_view(self, buf=None):
        buf = self.flat if buf is None else buf
        return buf[:self.layout.n_grid]

    def refresh_shadows(self, grid=True):
        """fp32 master -> fp16 table + MFMA fragments (after init / checkpoint load / optimizer step)."""
        L = self.layout
        if grid and L.n_grid and self.masters_stale:
            raise RuntimeError('Params.refresh_shadows: the fp32 table masters of the other ranks\' slices are stale (sharded optimizer); '
                               'call TrainEngine.sync_master() on every rank before anything rewrites or re-reads the parameters')
        if grid and L.n_grid:
            H.call('aln_cast_f16', H.ptr(self.flat), H.ptr(self.table16), L.n_grid, H.stream())
        fused = []
        for k, s in L.nets.items():
            if k in self.wide_w:
                o = L.offsets[k]
                for w, wt in zip(self.wide_w[k], self.wide_wt[k]):
                    w.copy_(self.flat[o:o + w.numel()].view_as(w))
                    H.call('aln_transpose_f16', H.ptr(w), w.shape[0], w.shape[1], H.ptr(wt), H.stream())
                    o += w.numel()
                if k == 'semf' and self.wide_wp is not None:
                    torch.index_select(self.wide_w[k][1], 1, self._kperm, out=self.wide_wp)
            else:
                fused.append(k)
        if fused:   # one launch for all fused heads
            n = len(fused)
            ws = (C.c_void_p * n)(*[self.flat.data_ptr() + 4 * L.offsets[k] for k in fused])
            ds = (C.c_void_p * n)(*[C.addressof(self.descs[k]) for k in fused])
            H.call('aln_mlp_repack_all', n, ws, ds, H.stream())


class Workspace:
    """Lazily allocated, shape-keyed device buffers (the caller owns all memory the kernels touch).

    ``generation`` counts (re)allocations: a captured hipGraph bakes in the raw pointers of the buffers it touched, so whoever
    replays one compares the generation it captured under with the current one (engine.GraphedStep).  The training engine owns
    a Workspace of its own (TrainEngine.ws): renders through the same pipeline use ``HipPipeline.ws`` and can never move the
    buffers a captured training step points into."""

    # Debug aid (tests/test_gpu_kernels.py, scripts/dev/stress_determinism.py): with guard_bytes > 0 every buffer is carved out of a
    # larger allocation with a canary band on either side, and new buffers are filled with NaN bytes (0xFF) instead of being left as
    # the allocator returned them -- check_guards() then shows a kernel that wrote outside its buffer, and a kernel that READS
    # scratch it did not write produces NaNs instead of plausible stale values.  Off (0) in the product: plain torch.empty.
    guard_bytes = 0
    CANARY = 0xA5

    def __init__(self, device):
        self.device, self.bufs, self.generation = device, {}, 0
        self._guarded = {}

    def _alloc(self, name, shape, dtype):
        if not Workspace.guard_bytes:
            return torch.empty(shape, dtype=dtype, device=self.device)
        g = int(Workspace.guard_bytes)
        assert g % 256 == 0
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (nbytes + 255) // 256 * 256
        raw = torch.full((g + pad + g,), Workspace.CANARY, dtype=torch.uint8, device=self.device)
        raw[g:g + nbytes] = 0xFF
        self._guarded[name] = (raw, g, nbytes)
        return raw[g:g + nbytes].view(dtype).reshape(shape)

    def check_guards(self):
        """Names of the buffers whose canary bands were overwritten (guard_bytes > 0 only)."""
        bad = []
        for name, (raw, g, nbytes) in self._guarded.items():
            lo, hi = raw[:g], raw[g + nbytes:]
            if bool((lo != Workspace.CANARY).any()) or bool((hi != Workspace.CANARY).any()):
                bad.append((name, int((lo != Workspace.CANARY).sum()), int((hi != Workspace.CANARY).sum())))
        return bad

    def get(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(name)
        if t is None or t[0] != key:
            self.bufs[name] = (key, self._alloc(name, shape, dtype))
            self.generation += 1
        return self.bufs[name][1]

    def scratch(self, name, nbytes):
        """Byte scratch that only ever grows (callers with varying sizes share one allocation)."""
        t = self.bufs.get(name)
        if t is None or t[1].numel() < nbytes:
            self.bufs[name] = (('scratch', name), self._alloc(name, (int(nbytes),), torch.uint8))
            self.generation += 1
        return self.bufs[name][1]


f16, f32_, i32_ = torch.float16, torch.float32, torch.int32


class OccupancyGrid:
    """Density grid + bitfield of the `cuda_ray` path (csrc/march.hip; hooks: autolabel/trainer.py:21-23,34-36,176).
    One level, G^3 cells over [-bound, bound]^3; `grid` / `bits` may be the model's registered buffers (they travel with
    state_dict like upstream's density_grid / density_bitfield)."""

    def __init__(self, device, G=128, max_steps=1024, samples=96, density_thresh=10.0, decay=0.95, update_interval=16,
                 grid=None, bits=None):
        self.G, self.max_steps, self.samples = int(G), int(max_steps), int(samples)
        self.density_thresh, self.decay, self.update_interval = float(density_thresh), float(decay), int(update_interval)
        n = self.G ** 3
        self.grid = grid if grid is not None else torch.zeros(n, dtype=f32_, device=device)
        self.bits = bits if bits is not None else torch.zeros((n + 31) // 32, dtype=i32_, device=device)
        assert self.grid.numel() == n and self.bits.numel() == (n + 31) // 32
        self.stats = torch.zeros(2, dtype=torch.int64, device=device)   # fixed-point sum and count of the grid mean (aln_grid_update)
        self.n_set = torch.zeros(1, dtype=i32_, device=device)
        self.updates = 0

    def occupancy(self):
        """Fraction of cells with their bit set (host sync: reporting only)."""
        return float(self.n_set.item()) / self.G ** 3


class HipPipeline:
    """render forward / backward as a fixed launch sequence (renderer semantics: oracle/nerf_oracle.py run())."""

    def __init__(self, layout, params, density_scale=1.0, min_near=0.2):
        H.require_gpu()
        self.L, self.P = layout, params
        self.ws = Workspace(params.device)
        self.density_scale, self.min_near = float(density_scale), float(min_near)
        self.found_inf = torch.zeros(1, dtype=i32_, device=params.device)
        self.phased_min_rows = 1 << 16   # hash-grid forward: level-phased from this many sample rows on
        self.tiled_enc_enabled = True    # (bench.py --no-tiled-enc: A/B against the plane buffers + assembly pass)
        # Rendering only.  Same-box A/B of the replayed training step (bench.py, two runs each): 2.245 / 2.241 M rays/s through planes +
        # assembly, 2.232 / 2.225 tiled (-0.6 %); marching step 5.68 / 5.70 vs 5.54 / 5.55 (-2.5 %); render through the grid 16.8 / 16.9 vs
        # 17.0 / 17.5 M rays/s (+1-4 %), dense render even.  The quarter-line stores cost the gather about what the assembly pass cost.
        self.tiled_enc_train = False
        # the tiled layout's 4-byte stores rely on the four levels of a 16-byte piece meeting in the XCD's L2: a phase writes rows x 4 B, an
        # eighth of it per XCD -- at the dense renderer's 8.4 M rows per launch that is more than the 4 MB L2 and the pieces leave for HBM
        # one level at a time (measured: dense render 4.8 -> 4.65 M rays/s; marching render, 2.1 M rows: 17.5 -> 18.3; training, 0.5 M: even)
        self.tiled_max_rows = 1 << 21     # rows per gather launch in tiled mode (density_rows splits larger passes into whole-ray pieces)
        # Training (round 6): the gather leaves the density head's input as PAIR PLANES -- every level's features as the coalesced 256-byte
        # wave stores of round 3's plane buffers, plus the frequency pairs and the ones -- and the 128-wide forward AND backward kernels read
        # the planes themselves (AlnMlpDesc.x_tiled = 2): no k_encode_assemble pass (2 x 22 us per step, 200 MB), no second buffer.
        self.planes_enc_train = True
        self.fold_dsigma = True          # (bench.py --no-fold-dsigma: A/B against the aln_assemble_grads pass)
        self.fold_color_in = True        # (bench.py --no-fold-color-in: A/B against the aln_build_color_in pass)
        # backward rebuilds hidden activations from the layer inputs (no h1/h2 saved in forward) when every fused head has a
        # recompute kernel; other shapes (e.g. 64-wide density / color nets) save them and use the generic backward kernels
        self.recompute = all(k in params.wide_w or H.lib().aln_mlp_has_recompute(s.in_pad, s.hidden, s.out_pad, s.n_hidden)
                             for k, s in layout.nets.items())
        self.occ = None     # OccupancyGrid: forward(..., march=True) then places the samples by marching (enable_marching)
        self.kernel_events = None  # bench.py: list of ((start, end) HIP events, kernel, tag) around the timed launches

    def _k(self, name, *args, tag=None):
        """H.call with optional HIP-event timing on the launch stream (bench.py sets kernel_events = [])."""
        if self.kernel_events is None:
            return H.call(name, *args)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        H.call(name, *args)
        ev[1].record()
        self.kernel_events.append((ev, name, tag))

    # ---- occupancy-grid marching (csrc/march.hip)
    def enable_marching(self, **kw):
        self.occ = OccupancyGrid(self.P.device, **kw)
        return self.occ

    def update_density_grid(self, step=0, seed=0x5EED, chunk=1 << 19, step_dev=None, ws=None):
        """NeRFRenderer.update_extra_state of upstream torch-ngp (the reference only carries the hooks: autolabel/trainer.py:21-23,
        34-36,176 call mark_untrained_grid; the refresh cadence -- every 16 steps -- is upstream's and this build's own spec):
        density of one jittered point per cell, grid = max(grid * decay, sigma * density_scale), bit = grid > min(mean, density_thresh)."""
        occ, e = self.occ, self.L.enc
        ws = self.ws if ws is None else ws
        n = occ.G ** 3
        sig = ws.get('occ_sigma', (n,), f32_)
        enc = ws.get('occ_enc', (chunk, e.enc_pad), f16)
        out = ws.get('occ_out', (chunk, 16), f16)
        planes = ws.scratch('enc_planes', int(e.grid.n_levels) * chunk * 4) if e.use_grid else None
        for a in range(0, n, chunk):
            rows = min(chunk, n - a)
            # the cell points (aln_grid_points' positions) are generated inside the encoding kernels: no [G^3, 3] buffer
            H.call('aln_encode_fwd_cells', C.byref(e), H.ptr(self.P.table16), occ.G, seed, step, H.ptr(step_dev), a, rows, H.ptr(planes),
                   H.ptr(enc), H.stream())
            H.call('aln_density_fwd', C.byref(self.P.descs['sigma']), H.ptr(enc), rows, None, None, H.ptr(out), H.ptr(sig[a:a + rows]), H.stream())
        H.call('aln_grid_update', H.ptr(occ.grid), H.ptr(sig), occ.G, occ.decay, self.density_scale, occ.density_thresh,
               H.ptr(occ.stats), H.ptr(occ.bits), H.ptr(occ.n_set), H.stream())
        occ.updates += 1

    def refresh_bitfield(self):
        """Bitfield from the grid as it is (after mark_untrained_grid)."""
        occ = self.occ
        H.call('aln_grid_update', H.ptr(occ.grid), None, occ.G, occ.decay, self.density_scale, occ.density_thresh,
               H.ptr(occ.stats), H.ptr(occ.bits), H.ptr(occ.n_set), H.stream())

    def recount_bitfield(self):
        """n_set from the bitfield as it is (a loaded checkpoint carries grid AND bitfield: the bits are kept, not re-derived)."""
        occ = self.occ
        H.call('aln_bitfield_count', H.ptr(occ.bits), occ.bits.numel(), H.ptr(occ.n_set), H.stream())

    def mark_untrained_grid(self, T_CW, intrinsics, size=None, z_near=0.0, sub=2):
        """Cells no camera sees get -1 (never occupied).  T_CW: [F,4,4] world -> OpenCV camera in the renderer's frame."""
        occ = self.occ
        T = torch.as_tensor(T_CW, dtype=f32_).to(self.P.device).contiguous()
        fx, fy, cx, cy = [float(v) for v in intrinsics]
        w, h = size if size is not None else (2 * cx + 1, 2 * cy + 1)
        H.call('aln_mark_untrained_grid', H.ptr(occ.grid), occ.G, self.L.enc.bound, H.ptr(T), T.shape[0], fx, fy, cx, cy, float(w),
               float(h), float(z_near), int(sub), H.stream())
        torch.cuda.current_stream().synchronize()   # T is a temporary
        self.refresh_bitfield()

    def binned_record_count(self, M, ws=None):
        """PAIR records (12 bytes: the two x-neighbour corners of a cell) the last binned hash-grid backward over M sample rows streamed
        through HBM (sum of the per-(level, slice, tile) descriptor counts; written once by phase 1 and read once by phase 2).  Host
        sync: reporting only."""
        t = (self.ws if ws is None else ws).bufs.get('enc_bwd_bins')
        if t is None:
            return None
        tile = int(H.lib().aln_encode_bwd_binned_tile_rows())
        nl, nt = int(self.L.enc.grid.n_levels), (M + tile - 1) // tile
        pool = int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(self.L.enc), M)) - nl * 64 * nt * 4   # (the descriptors close the workspace)
        desc = t[1][pool:pool + nl * 64 * nt * 4].view(torch.int32)
        return int(((desc >> 13) & 0x3FFF).sum().item())

    # ---- wide semantic heads on the hand-written MFMA GEMMs of wide.hip (models.py:248-256 at LSeg width)
    def _nt(self, M, N, w, y, a1=None, K1=0, relu1=0, geo=None, relu=0, mask=None, add=None, watch=False, tag=None):
        self._k('aln_wide_nt', H.ptr(a1), a1.shape[1] if a1 is not None else 0, K1, relu1, H.ptr(geo), self.L.G, M, N, H.ptr(w), w.shape[1],
                H.ptr(y), y.shape[1], relu, H.ptr(mask), mask.shape[1] if mask is not None else 0, H.ptr(add),
                add.shape[1] if add is not None else 0, H.ptr(self.found_inf) if watch else None, H.stream(),
                tag=('wide', M, N * (K1 + (16 if geo is not None else 0))))

    def _tn(self, M, N, g, dw_off, ldw, a1=None, K1=0, relu1=0, geo=None, tag=None):
        K = K1 + (16 if geo is not None else 0)
        scratch = self._tn_ws.scratch('wide_tn_slabs', int(H.lib().aln_wide_tn_ws_bytes(M, N, K)))   # per-block partial sums, reduced in a fixed order
        self._k('aln_wide_tn', H.ptr(g), g.shape[1], H.ptr(a1), a1.shape[1] if a1 is not None else 0, K1, relu1, H.ptr(geo), self.L.G, M, N,
                C.c_void_p(self.P.grad.data_ptr() + 4 * dw_off), ldw, H.ptr(scratch), H.stream(), tag=('wide', M, N * K))

    # generated first layer of semantic_features (wide.hip): h1 = relu([geo_feat, 1] W0^T) is recomputed by whoever needs it
    def _gen(self):
        return self.P.wide_wp is not None

    def _nt_gen(self, M, y, sout, relu=1, w_row=None, tile_sums=None, tag=None):
        fs, W = self.L.nets['semf'], self.P.wide_w['semf']
        self._k('aln_wide_nt_gen', H.ptr(sout), self.L.G, H.ptr(W[0]), M, fs.hidden, fs.hidden, H.ptr(self.P.wide_wp), fs.hidden, H.ptr(y), y.shape[1], relu,
                H.ptr(w_row), H.ptr(tile_sums), None, H.stream(), tag=('wide', M, fs.hidden * (fs.hidden + 16)))

    def _nt_maskgen(self, M, wt, y, a1, sout, tag=None):
        fs = self.L.nets['semf']
        self._k('aln_wide_nt_maskgen', H.ptr(a1), a1.shape[1], M, fs.hidden, fs.hidden, H.ptr(wt), wt.shape[1], H.ptr(y), y.shape[1], H.ptr(sout), self.L.G,
                H.ptr(self.P.wide_w['semf'][0]), H.ptr(self.found_inf), H.stream(), tag=('wide', M, fs.hidden * (fs.hidden + 16)))

    def _tn_gen(self, M, g, dw_off, sout, tag=None):
        fs = self.L.nets['semf']
        scratch = self._tn_ws.scratch('wide_tn_slabs', int(H.lib().aln_wide_tn_ws_bytes(M, fs.hidden, fs.hidden)))
        self._k('aln_wide_tn_gen', H.ptr(g), g.shape[1], H.ptr(sout), self.L.G, H.ptr(self.P.wide_w['semf'][0]), M, fs.hidden, fs.hidden,
                C.c_void_p(self.P.grad.data_ptr() + 4 * dw_off), fs.hidden, H.ptr(scratch), H.stream(), tag=('wide', M, fs.hidden * (fs.hidden + 16)))

    def _tn_din(self, M, dh1, d_fin, sout, dw_off, tag=None):
        """d_fin = dh1 W0 and dW0 += dh1^T [geo_feat, 1] in one pass over dh1 (aln_wide_tn_din)."""
        fs = self.L.nets['semf']
        scratch = self._tn_ws.scratch('wide_tn_slabs', int(H.lib().aln_wide_tn_din_ws_bytes(M, fs.hidden)))
        wt = self.P.wide_wt['semf'][0]
        self._k('aln_wide_tn_din', H.ptr(dh1), dh1.shape[1], H.ptr(sout), self.L.G, H.ptr(wt), wt.shape[1], M, fs.hidden,
                C.c_void_p(self.P.grad.data_ptr() + 4 * dw_off), fs.in_pad, H.ptr(scratch), H.ptr(d_fin), H.ptr(self.found_inf), H.stream(),
                tag=('wide', M, fs.hidden * 32))

    def _din_fused(self):
        fs = self.L.nets['semf']
        return fs.in_pad == 16 and fs.hidden <= 512 and fs.hidden % 32 == 0

    def wide_sem_fwd(self, sout, M, bufs):
        """f = semantic_features([geo, 1]); logits = semantic_out([relu(f), geo, 1]) for M rows of the density head's output.
        bufs: callable name, shape -> fp16 buffer.  Returns (logits [M, Cpad], f [M, D], saved activations)."""
        L = self.L
        fs, os_ = L.nets['semf'], L.nets['semo']
        Wf, Wo = self.P.wide_w['semf'], self.P.wide_w['semo']
        fl = ('sem', M)
        h2 = bufs('wide_h2', (M, fs.hidden))
        feat, ho = bufs('feat', (M, fs.out_pad)), bufs('wide_ho', (M, os_.hidden))
        logits = bufs('logits', (M, os_.out_pad))
        if self._gen():    # layers 1 + 2 in one launch, h1 never stored
            h1 = None
            self._nt_gen(M, h2, sout, tag=fl)
        else:
            h1 = bufs('wide_h1', (M, fs.hidden))
            self._nt(M, fs.hidden, Wf[0], h1, geo=sout, relu=1, tag=fl)
            self._nt(M, fs.hidden, Wf[1], h2, a1=h1, K1=fs.hidden, relu=1, tag=fl)
        self._nt(M, fs.out_pad, Wf[2], feat, a1=h2, K1=fs.hidden, tag=fl)
        self._nt(M, os_.hidden, Wo[0], ho, a1=feat, K1=L.D, relu1=1, geo=sout, relu=1, tag=fl)
        self._nt(M, os_.out_pad, Wo[1], logits, a1=ho, K1=os_.hidden, tag=fl)
        return logits, feat, (h1, h2, ho)

    def wide_sem_bwd(self, sout, M, feat, saved, d_logits, d_feat, bufs):
        """Backward of wide_sem_fwd: weight gradients into P.grad, returns (d semf_in [M,16], d geo-part of semo_in [M,16]).
        d_feat is updated in place with the gradient arriving through relu(f) -> semantic_out."""
        L = self.L
        fs, os_ = L.nets['semf'], L.nets['semo']
        Wft, Wot = self.P.wide_wt['semf'], self.P.wide_wt['semo']
        h1, h2, ho = saved
        bl = ('sem', M)
        of, oo = L.offsets['semf'], L.offsets['semo']
        sf, so = fs.shapes, os_.shapes
        # ---- semantic_out: logits = W2 relu(W1 [relu f, geo, 1])
        dho = bufs('wide_dho', (M, os_.hidden))
        self._nt(M, os_.hidden, Wot[1], dho, a1=d_logits, K1=os_.out_pad, mask=ho, watch=True, tag=bl)
        self._tn(M, os_.out_pad, d_logits, oo + so[0][0] * so[0][1], os_.hidden, a1=ho, K1=os_.hidden, tag=bl)
        self._tn(M, os_.hidden, dho, oo, os_.in_pad, a1=feat, K1=L.D, relu1=1, geo=sout, tag=bl)
        d_ogeo = bufs('wide_d_ogeo', (M, 16))
        self._nt(M, L.D, Wot[0][:L.D], d_feat, a1=dho, K1=os_.hidden, mask=feat, add=d_feat, watch=True, tag=bl)
        self._nt(M, 16, Wot[0][L.D:L.D + 16], d_ogeo, a1=dho, K1=os_.hidden, watch=True, tag=bl)
        # ---- semantic_features: f = W3 relu(W2 relu(W1 [geo, 1]))
        dh2, dh1 = bufs('wide_dh2', (M, fs.hidden)), bufs('wide_dh1', (M, fs.hidden))
        n0, n1 = sf[0][0] * sf[0][1], sf[1][0] * sf[1][1]
        self._nt(M, fs.hidden, Wft[2], dh2, a1=d_feat, K1=fs.out_pad, mask=h2, watch=True, tag=bl)
        self._tn(M, fs.out_pad, d_feat, of + n0 + n1, fs.hidden, a1=h2, K1=fs.hidden, tag=bl)
        if h1 is None:
            self._nt_maskgen(M, Wft[1], dh1, dh2, sout, tag=bl)
            self._tn_gen(M, dh2, of + n0, sout, tag=bl)
        else:
            self._nt(M, fs.hidden, Wft[1], dh1, a1=dh2, K1=fs.hidden, mask=h1, watch=True, tag=bl)
            self._tn(M, fs.hidden, dh2, of + n0, fs.hidden, a1=h1, K1=fs.hidden, tag=bl)
        d_fin = bufs('d_semf_in', (M, fs.in_pad))
        if self._din_fused():
            self._tn_din(M, dh1, d_fin, sout, of, tag=bl)
        else:
            self._nt(M, fs.in_pad, Wft[0], d_fin, a1=dh1, K1=fs.hidden, watch=True, tag=bl)
            self._tn(M, fs.hidden, dh1, of, fs.in_pad, geo=sout, tag=bl)
        return d_fin, d_ogeo

    # ---- point queries (models.py:175-188, 190-220, 248-256)
    def tiled_enc(self, rows_per_pass, train):
        """True when the density head's input rows are written straight in the tiled layout (no plane buffers, no assembly pass): the
        level-phased gather for every pass of the launch sequence, the 128-wide kernels as the only readers, and pass boundaries on
        whole 32-row tiles."""
        if train and not self.tiled_enc_train:
            return False
        save = train and not self.recompute
        return (self.tiled_enc_enabled and self.P.desc_sigma_tiled is not None and bool(self.L.enc.use_grid) and not save and
                all(self.phased_min_rows <= r and r % 32 == 0 for r in rows_per_pass))

    def planes_enc(self, rows_per_pass, train):
        """True when the training step's density head reads pair planes (see planes_enc_train): level-phased gathers for every pass,
        recompute backward (nobody else reads the rows), whole 32-row tiles per pass."""
        return (train and self.planes_enc_train and self.P.desc_sigma_planes is not None and bool(self.L.enc.use_grid) and self.recompute and
                sum(rows_per_pass) % 4 == 0 and all(self.phased_min_rows <= r and r % 32 == 0 for r in rows_per_pass))

    def density_rows(self, rows, rays_o, rays_d, z, xyz, stride, enc, h1, h2, out, sigma, train, ws=None, tiled=False, planes=None):
        e, s = self.L.enc, self.L.nets['sigma']
        ws = self.ws if ws is None else ws
        if planes is not None:    # (whole plane buffer [enc_pad / 2, pitch] words, first row of this pass)
            buf, pitch, a = planes
            at = C.c_void_p(buf.data_ptr() + 4 * a)
            self._k('aln_encode_fwd_planes', C.byref(e), H.ptr(self.P.table16), H.ptr(rays_o), H.ptr(rays_d), H.ptr(z), H.ptr(xyz), rows, stride,
                    at, pitch, H.stream(), tag=('enc_fwd', rows))
            d = self.P.desc_sigma_planes
            d.x_pitch = pitch
            self._k('aln_density_fwd', C.byref(d), at, rows, None, None, H.ptr(out), H.ptr(sigma), H.stream(), tag=('sigma', rows))
            return
        if tiled:
            # (launches of at most tiled_max_rows rows, whole rays each: see tiled_max_rows)
            per = max(stride, self.tiled_max_rows // (32 * stride) * (32 * stride)) if xyz is None else self.tiled_max_rows // 32 * 32
            for a in range(0, rows, per):
                n = min(per, rows - a)
                ra = a // stride if xyz is None else 0
                self._k('aln_encode_fwd_phased', C.byref(e), H.ptr(self.P.table16), H.ptr(rays_o[ra:]) if xyz is None else None,
                        H.ptr(rays_d[ra:]) if xyz is None else None, H.ptr(z[a:]) if xyz is None else None, H.ptr(xyz[a:]) if xyz is not None else None,
                        n, stride, None, H.ptr(enc[a:]), H.stream(), tag=('enc_fwd', n))
            self._k('aln_density_fwd', C.byref(self.P.desc_sigma_tiled), H.ptr(enc), rows, None, None, H.ptr(out), H.ptr(sigma), H.stream(),
                    tag=('sigma', rows))
            return
        if e.use_grid and rows >= self.phased_min_rows:
            # large batches: level-phased gathers (tables in flight stay L2-resident) + streaming row assembly
            planes = ws.scratch('enc_planes', int(e.grid.n_levels) * rows * 4)
            self._k('aln_encode_fwd_phased', C.byref(e), H.ptr(self.P.table16), H.ptr(rays_o), H.ptr(rays_d), H.ptr(z), H.ptr(xyz),
                    rows, stride, H.ptr(planes), H.ptr(enc), H.stream(), tag=('enc_fwd', rows))
        else:
            H.call('aln_encode_fwd', C.byref(e), H.ptr(self.P.table16), H.ptr(rays_o), H.ptr(rays_d), H.ptr(z), H.ptr(xyz), rows,
                   stride, H.ptr(enc), H.stream())
        save = train and not self.recompute
        self._k('aln_density_fwd', C.byref(self.P.descs['sigma']), H.ptr(enc), rows, H.ptr(h1) if save else None,
                H.ptr(h2) if save else None, H.ptr(out), H.ptr(sigma), H.stream(), tag=('sigma', rows))

    def forward(self, rays_o, rays_d, norms, S1, S2, perturb, train, seed=0, step=0, noise=None, u=None,
                want_semantic=True, bg=1.0, step_dev=None, march=False, ws=None, sem_linear=False):
        """``sem_linear`` (training of the wide / LSeg heads with semantic_weight = 0, scripts/ros/node.py:166-176): the class
        logits carry no loss, so semantic_out is skipped, and the last layer of semantic_features -- linear, no activation -- is
        applied ONCE PER RAY to the composited hidden activation instead of once per sample: sum_s w_s (W3 h2_s) = W3 (sum_s w_s h2_s)
        (autolabel/models.py:123, 248-256).  out['semantic'] is then zeros."""
        L, P = self.L, self.P
        ws = self.ws if ws is None else ws     # (the training engine passes its own: see Workspace)
        N = rays_o.shape[0]
        if march:   # S1 rows per ray placed inside occupied cells, no importance pass (noise: [N] per-ray jitter)
            assert self.occ is not None, 'forward(march=True) needs enable_marching()'
            S2 = 0
        S, M = S1 + S2, N * (S1 + S2)
        M1 = N * S1
        e, nets = L.enc, L.nets
        dev = rays_o.device
        g = lambda n, shp, dt: ws.get(n, shp, dt)
        ws.serial = getattr(ws, 'serial', 0) + 1   # every forward rewrites the name-keyed buffers of its workspace
        c = dict(N=N, S1=S1, S2=S2, M=M, train=train, want_semantic=want_semantic, bg=float(bg),
                 rays_o=rays_o, rays_d=rays_d, norms=norms, serial=ws.serial, ws=ws)
        save = train and not self.recompute
        c['nears'], c['fars'] = g('nears', (N,), f32_), g('fars', (N,), f32_)
        z = c['z'] = g('z', (M,), f32_)
        enc = c['enc'] = g('enc', (M, e.enc_pad), f16)
        hs = nets['sigma'].hidden
        h1 = c['h1'] = g('h1', (M if save else 1, hs), f16)
        h2 = c['h2'] = g('h2', (M if save else 1, hs), f16)
        sout = c['sigma_out'] = g('sigma_out', (M, 16), f16)
        sigma = c['sigma'] = g('sigma', (M,), f32_)
        delta_in = None
        if march:
            occ = self.occ
            delta_in = c['delta_in'] = g('delta_in', (M,), f32_)
            H.call('aln_march_rays', H.ptr(rays_o), H.ptr(rays_d), N, S1, e.bound, self.min_near, H.ptr(occ.bits), occ.G, occ.max_steps,
                   int(perturb), seed, step, H.ptr(step_dev), H.ptr(noise), H.ptr(c['nears']), H.ptr(c['fars']), H.ptr(z),
                   H.ptr(delta_in), None, H.stream())
        else:
            H.call('aln_sample_coarse', H.ptr(rays_o), H.ptr(rays_d), N, S1, e.bound, self.min_near, int(perturb), seed, step,
                   H.ptr(noise), H.ptr(c['nears']), H.ptr(c['fars']), H.ptr(z), H.ptr(step_dev), H.stream())
        passes = [M1] + ([N * S2] if S2 > 0 else [])
        c['enc_planes'] = self.planes_enc(passes, train)
        tiled = c['enc_tiled'] = not c['enc_planes'] and self.tiled_enc(passes, train)
        self.density_rows(M1, rays_o, rays_d, z, None, S1, enc, h1, h2, sout, sigma, train, ws=ws, tiled=tiled,
                          planes=(enc, M, 0) if c['enc_planes'] else None)
        if S2 > 0:
            zf = z[M1:]
            H.call('aln_sample_fine', H.ptr(z), H.ptr(sigma), H.ptr(c['nears']), H.ptr(c['fars']), N, S1, S2,
                   self.density_scale, int(perturb), seed, step, H.ptr(u), H.ptr(zf), H.ptr(step_dev), H.stream())
            self.density_rows(N * S2, rays_o, rays_d, zf, None, S2, enc[M1:], h1[M1:] if save else h1,
                              h2[M1:] if save else h2, sout[M1:], sigma[M1:], train, ws=ws, tiled=tiled,
                              planes=(enc, M, M1) if c['enc_planes'] else None)
        perm = c['perm'] = g('perm', (N, S), torch.int16)
        w_row, T_row, d_row = g('w_row', (M,), f32_), g('T_row', (M,), f32_), g('delta_row', (M,), f32_)
        c.update(w_row=w_row, T_row=T_row, delta_row=d_row)
        out = {k: torch.empty(shp, dtype=f32_, device=dev) for k, shp in
               [('weights_sum', (N,)), ('depth', (N,)), ('depth_variance', (N,)), ('coordinates_map', (N, 3)),
                ('image', (N, 3))]}
        H.call('aln_composite_fwd', H.ptr(rays_o), H.ptr(rays_d), H.ptr(norms), H.ptr(c['nears']), H.ptr(c['fars']), H.ptr(z),
               H.ptr(sigma), N, S1, S2, e.bound, self.density_scale, H.ptr(perm), H.ptr(w_row), H.ptr(T_row), H.ptr(d_row),
               H.ptr(out['weights_sum']), H.ptr(out['depth']), H.ptr(out['depth_variance']), H.ptr(out['coordinates_map']),
               H.ptr(delta_in), H.stream())
        # color head on live samples only (models.py:195-203)
        n_live, live_idx, cidx = g('n_live', (1,), i32_), g('live_idx', (M,), i32_), g('cidx_row', (M,), i32_)
        c.update(n_live=n_live, live_idx=live_idx, cidx_row=cidx)
        chunk_ws = g('compact_ws', (max(int(H.lib().aln_compact_live_ws_ints(M)), 1),), i32_)
        cs = nets['color']
        cin = c['color_in'] = g('color_in', (M, cs.in_pad), f16)
        build_cin = train or cs.in_pad != 32
        if build_cin and self.fold_color_in:   # (the input rows of the colour head leave the compaction's second pass: no k_build_color_in launch)
            H.call('aln_compact_live_color_in', H.ptr(w_row), M, 1e-4, H.ptr(n_live), H.ptr(live_idx), H.ptr(cidx), H.ptr(chunk_ws),
                   H.ptr(rays_d), None, N, S1, S2, H.ptr(sout), L.G, cs.in_pad, H.ptr(cin), H.stream())
        else:
            H.call('aln_compact_live', H.ptr(w_row), M, 1e-4, H.ptr(n_live), H.ptr(live_idx), H.ptr(cidx), H.ptr(chunk_ws), H.stream())
        ch1 = c['ch1'] = g('ch1', (M if save else 1, cs.hidden), f16)
        ch2 = c['ch2'] = g('ch2', (M if save else 1, cs.hidden), f16)
        cout = c['color_out'] = g('color_out', (M, cs.out_pad), f16)
        if build_cin:
            if not self.fold_color_in:
                H.call('aln_build_color_in', H.ptr(live_idx), H.ptr(n_live), M, H.ptr(rays_d), None, N, S1, S2, H.ptr(sout), L.G,
                       cs.in_pad, H.ptr(cin), H.stream())
            self._k('aln_mlp_fwd', C.byref(P.descs['color']), H.ptr(cin), M, H.ptr(n_live), H.ptr(ch1) if save else None,
                    H.ptr(ch2) if save else None, H.ptr(cout), H.stream(), tag=('color', n_live))
        else:   # inference: the input rows are built inside the kernel (no color_in round trip through HBM)
            H.call('aln_color_fwd', C.byref(P.descs['color']), H.ptr(live_idx), H.ptr(n_live), M, H.ptr(rays_d), None, N, S1, S2,
                   H.ptr(sout), L.G, H.ptr(cout), H.stream())
        logits = feat = None
        if want_semantic:
            fs, os_ = nets['semf'], nets['semo']
            c['sem_fused'] = self.recompute and not L.sem_wide
            c['sem_wide'] = L.sem_wide
            c['sem_linear'] = bool(sem_linear and L.sem_wide and train)
            if c['sem_linear']:
                Wf = P.wide_w['semf']
                h2 = g('wide_h2', (M, fs.hidden), f16)
                if self._gen():
                    h1 = None
                    # the composited hidden activation comes out of the GEMM's epilogue as per-tile weighted sums: no second pass over h2
                    if S1 % 32 == 0 and S2 % 32 == 0:
                        c['h2_tile_sums'] = g('wide_h2_tile_sums', (M // 32, fs.hidden), f32_)
                    self._nt_gen(M, h2, sout, w_row=w_row if 'h2_tile_sums' in c else None, tile_sums=c.get('h2_tile_sums'), tag=('sem', M))
                else:
                    h1 = g('wide_h1', (M, fs.hidden), f16)
                    self._nt(M, fs.hidden, Wf[0], h1, geo=sout, relu=1, tag=('sem', M))
                    self._nt(M, fs.hidden, Wf[1], h2, a1=h1, K1=fs.hidden, relu=1, tag=('sem', M))
                c['wide_saved'] = (h1, h2, None)
                c['feat'], c['logits'] = h2, None      # what the compositing kernels sum / differentiate: the hidden activation
            elif c['sem_wide']:
                logits, feat, c['wide_saved'] = self.wide_sem_fwd(sout, M, lambda n, shp: g(n, shp, f16))
                c['feat'], c['logits'] = feat, logits
            elif c['sem_fused'] and S1 % 32 == 0 and S2 % 32 == 0 and \
                    H.lib().aln_sem_heads_bwd_slabs(C.byref(P.descs['semf']), C.byref(P.descs['semo']), M, L.D, L.G) > 0:
                # neither f nor the logits are stored (all anybody needs of them are the per-ray weighted sums): the forward leaves
                # the weighted sums of every 32-row tile (one ray each); in the training step the one-kernel backward recomputes
                # both and hands the compositing backward its dot products.  Rendering takes the same path (512 / 128 rows per ray).
                c['sem_sums'] = True
                feat = logits = c['feat'] = c['logits'] = None
                tile_sums = g('sem_tile_sums', (M // 32, 96), f32_)
                self._k('aln_sem_heads_fwd_sums', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(sout), M, L.D, L.G,
                        H.ptr(w_row), H.ptr(tile_sums), H.stream(), tag=('sem', M))
            elif c['sem_fused']:  # inputs are built inside the kernels from sigma_out / f (no semf_in / semo_in tensors)
                feat = c['feat'] = g('feat', (M, fs.out_pad), f16)
                logits = c['logits'] = g('logits', (M, os_.out_pad), f16)
                self._k('aln_sem_heads_fwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(sout), M, L.D, L.G,
                        H.ptr(feat), H.ptr(logits), H.stream(), tag=('sem', M))
            else:
                fin = c['semf_in'] = g('semf_in', (M, fs.in_pad), f16)
                oin = c['semo_in'] = g('semo_in', (M, os_.in_pad), f16)
                H.call('aln_build_sem_in', H.ptr(sout), None, M, L.D, L.G, fs.in_pad, os_.in_pad, H.ptr(fin), None, H.stream())
                fh1 = c['fh1'] = g('fh1', (M if save else 1, fs.hidden), f16)
                fh2 = c['fh2'] = g('fh2', (M if save else 1, fs.hidden), f16)
                feat = c['feat'] = g('feat', (M, fs.out_pad), f16)
                H.call('aln_mlp_fwd', C.byref(P.descs['semf']), H.ptr(fin), M, None, H.ptr(fh1) if save else None,
                       H.ptr(fh2) if save else None, H.ptr(feat), H.stream())
                H.call('aln_build_sem_in', H.ptr(sout), H.ptr(feat), M, L.D, L.G, fs.in_pad, os_.in_pad, None, H.ptr(oin), H.stream())
                oh1 = c['oh1'] = g('oh1', (M if save else 1, os_.hidden), f16)
                logits = c['logits'] = g('logits', (M, os_.out_pad), f16)
                H.call('aln_mlp_fwd', C.byref(P.descs['semo']), H.ptr(oin), M, None, H.ptr(oh1) if save else None, None,
                       H.ptr(logits), H.stream())
            out['semantic'] = torch.empty((N, L.C), dtype=f32_, device=dev)
            out['semantic_features'] = torch.empty((N, L.D), dtype=f32_, device=dev)
        if c.get('sem_linear'):
            # composite the hidden activation per ray, then ONE [N, hidden] x [hidden, D] GEMM for the whole batch
            fs = nets['semf']
            h2r = c['h2_ray32'] = g('h2_ray32', (N, fs.hidden), f32_)
            h2r16 = c['h2_ray16'] = g('h2_ray16', (N, fs.hidden), f16)
            f16r = g('feat_ray16', (N, fs.out_pad), f16)
            if 'h2_tile_sums' in c:
                H.call('aln_composite_out_featsums', H.ptr(w_row), H.ptr(cidx), H.ptr(cout), H.ptr(out['weights_sum']), N, S1, S2, fs.hidden, float(bg),
                       H.ptr(out['image']), H.ptr(h2r), H.ptr(c['h2_tile_sums']), H.stream())
            else:
                H.call('aln_composite_out', H.ptr(w_row), H.ptr(cidx), H.ptr(cout), None, H.ptr(c['feat']), H.ptr(out['weights_sum']),
                       N, S1, S2, L.C, L.Cpad, fs.hidden, float(bg), H.ptr(out['image']), None, H.ptr(h2r), None, H.stream())
            H.call('aln_cast_f16', H.ptr(h2r), H.ptr(h2r16), h2r.numel(), H.stream())
            self._nt(N, fs.out_pad, P.wide_w['semf'][2], f16r, a1=h2r16, K1=fs.hidden, tag=('sem', N))
            H.call('aln_cast_f32', H.ptr(f16r), H.ptr(out['semantic_features']), f16r.numel(), H.stream())
            out['semantic'].zero_()
            return out, c
        H.call('aln_composite_out', H.ptr(w_row), H.ptr(cidx), H.ptr(cout), H.ptr(logits), H.ptr(feat), H.ptr(out['weights_sum']),
               N, S1, S2, L.C, L.Cpad, L.D, float(bg), H.ptr(out['image']), H.ptr(out.get('semantic')),
               H.ptr(out.get('semantic_features')), H.ptr(tile_sums) if c.get('sem_sums') else None, H.stream())
        return out, c

    def backward(self, c, g_image, g_depth, g_sem=None, g_feat=None, level_groups=None, on_grad_ready=None, scatter_flag=None, grid_adam=None,
                 grid_wire=None, wire_mul=1.0):
        """Accumulate d(loss)/d(params) into P.grad from per-ray output gradients (fp32, already loss-scaled).

        Data-parallel callers pass ``level_groups`` = [(lo, hi), ...] and ``on_grad_ready``: the hash-grid scatter then runs
        one group of levels at a time and the callback fires as soon as a part of P.grad is final -- ``('mlp', a, b)`` for
        the MLP block (+ overflow flag) before the scatter starts, ``('grid', a, b)`` after each group (flat offsets) -- so
        its all-reduce overlaps the remaining scatter.  ``scatter_flag`` (int32[1]): the hash-grid scatter raises this word instead
        of ``found_inf`` (a data-parallel caller ships ``found_inf`` with the MLP bucket while the scatter is still running).
        ``grid_adam`` (hip.AlnAdamFuse): the scatter's second phase takes the optimizer step for the table itself instead of adding
        the table's gradient to P.grad (single-GPU training: TrainEngine.step).
        ``grid_wire`` (fp16 [n_grid]) with ``wire_mul`` (data parallelism, fp16 on the wire): the table's gradient leaves the scatter as the
        exchange's payload, fp16(sum * wire_mul) -- nothing of it goes to P.grad (aln_encode_bwd_binned_wire)."""
        L, P, ws = self.L, self.P, c['ws']
        self._tn_ws = ws      # scratch of the wide heads' weight-gradient GEMMs (_tn)
        assert c['train'], 'backward needs a forward(train=True) context'
        if c.get('serial') != getattr(ws, 'serial', None):
            raise RuntimeError('HipPipeline.backward: another forward() ran through the same workspace since the context was created; '
                               'the intermediates live in its name-keyed buffers and have been overwritten (call backward before the '
                               'next render, e.g. one render() per loss, or render previews under a second model)')
        N, S1, S2, M = c['N'], c['S1'], c['S2'], c['M']
        M1 = N * S1
        nets, e = L.nets, L.enc
        g = lambda n, shp, dt: ws.get(n, shp, dt)
        sem = c['want_semantic'] and g_sem is not None
        lin = sem and c.get('sem_linear', False)
        fi = H.ptr(self.found_inf)
        gp = lambda k: C.c_void_p(P.grad.data_ptr() + 4 * L.offsets[k])
        d_h0 = g('d_h0', (M,), f32_)
        cs = nets['color']
        d_cout = g('d_color_out', (M, cs.out_pad), f16)
        sem_fused = sem and c.get('sem_fused', False)
        d_logits = g('d_logits', (M, L.Cpad), f16) if sem and not sem_fused and not c.get('sem_linear') else None
        d_feat = g('d_feat', (M, L.D), f16) if sem and not sem_fused and not c.get('sem_linear') else None
        if sem and g_feat is None:
            g_feat = torch.zeros((N, L.D), dtype=f32_, device=g_image.device)
        if lin:
            # per ray: dW3 += dF^T H2 ; dH2 = dF W3 ; per sample (inside the compositing backward): d h2_s = w_s relu'(h2_s) dH2[ray],
            # d w_s += <h2_s, dH2[ray]>
            fs = nets['semf']
            of = L.offsets['semf']
            n0, n1 = fs.shapes[0][0] * fs.shapes[0][1], fs.shapes[1][0] * fs.shapes[1][1]
            dF16 = g('d_feat_ray16', (N, fs.out_pad), f16)
            dH16 = g('d_h2_ray16', (N, fs.hidden), f16)
            dH32 = g('d_h2_ray32', (N, fs.hidden), f32_)
            H.call('aln_cast_f16', H.ptr(g_feat), H.ptr(dF16), g_feat.numel(), H.stream())
            self._tn(N, fs.out_pad, dF16, of + n0 + n1, fs.hidden, a1=c['h2_ray16'], K1=fs.hidden, tag=('sem', N))
            self._nt(N, fs.hidden, P.wide_wt['semf'][2], dH16, a1=dF16, K1=fs.out_pad, watch=True, tag=('sem', N))
            H.call('aln_cast_f32', H.ptr(dH16), H.ptr(dH32), dH16.numel(), H.stream())
            d_feat = g('wide_dh2', (M, fs.hidden), f16)    # = d h2 (masked): what the remaining layers back-propagate
            H.call('aln_composite_bwd', H.ptr(c['norms']), H.ptr(c['z']), H.ptr(c['sigma']), H.ptr(c['perm']), H.ptr(c['w_row']),
                   H.ptr(c['T_row']), H.ptr(c['delta_row']), H.ptr(c['cidx_row']), H.ptr(c['color_out']), None, H.ptr(c['feat']),
                   H.ptr(c['sigma_out']), H.ptr(g_image), H.ptr(g_depth), None, H.ptr(dH32), N, S1, S2, L.C, L.Cpad, fs.hidden,
                   c['bg'], self.density_scale, H.ptr(d_h0), H.ptr(d_cout), None, H.ptr(d_feat), 1, None, fi, H.stream())
        else:
            dots = d_fin_pair = None
            if sem_fused and c.get('sem_sums'):
                # the semantic pair goes FIRST: it needs only the compositing weights and the per-ray output gradients, and its
                # per-row <logits, g_sem> + <f, g_feat> is the semantic outputs' share of dL/dw the compositing backward wants
                fs = nets['semf']
                dots, d_fin_pair = g('sem_dots', (M,), f32_), g('d_semf_in', (M, fs.in_pad), f16)
                self._k('aln_sem_heads_bwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(c['sigma_out']), None,
                        H.ptr(c['w_row']), H.ptr(g_sem), H.ptr(g_feat), N, S1, S2, L.C, M, L.D, L.G, None, H.ptr(d_fin_pair),
                        gp('semf'), gp('semo'), 1, H.ptr(dots), fi, H.stream(), tag=('sem', M))
            elif sem and c.get('sem_sums'):
                raise RuntimeError('HipPipeline.backward: the forward kept no semantic rows (sem_sums) but the one-kernel backward is not available')
            H.call('aln_composite_bwd', H.ptr(c['norms']), H.ptr(c['z']), H.ptr(c['sigma']), H.ptr(c['perm']), H.ptr(c['w_row']),
                   H.ptr(c['T_row']), H.ptr(c['delta_row']), H.ptr(c['cidx_row']), H.ptr(c['color_out']),
                   H.ptr(c['logits']) if sem and dots is None else None, H.ptr(c['feat']) if sem and dots is None else None, H.ptr(c['sigma_out']),
                   H.ptr(g_image), H.ptr(g_depth), H.ptr(g_sem) if sem and dots is None else None, H.ptr(g_feat) if sem and dots is None else None,
                   N, S1, S2, L.C, L.Cpad, L.D, c['bg'], self.density_scale, H.ptr(d_h0), H.ptr(d_cout), H.ptr(d_logits), H.ptr(d_feat), 0,
                   H.ptr(dots), fi, H.stream())
        # color head
        rc = self.recompute
        hp = (lambda t: None) if rc else H.ptr   # saved activations are not passed on the recompute path
        dA1, dA2 = (None, None) if rc else (g('dA1', (M, 128), f16), g('dA2', (M, 128), f16))
        d_cin = g('d_color_in', (M, cs.in_pad), f16)
        self._k('aln_mlp_bwd', C.byref(P.descs['color']), H.ptr(c['color_in']), hp(c['ch1']), hp(c['ch2']), H.ptr(d_cout),
                M, H.ptr(c['n_live']), H.ptr(dA1), H.ptr(dA2), H.ptr(d_cin), gp('color'), fi, H.stream(), tag=('color', c['n_live']))
        d_fin = d_oin = None
        if sem_fused and c.get('sem_sums'):
            d_fin = d_fin_pair       # (ran before the compositing backward, above)
        elif sem_fused:
            fs, os_ = nets['semf'], nets['semo']
            d_oin, d_fin = g('d_semo_in', (M, os_.in_pad), f16), g('d_semf_in', (M, fs.in_pad), f16)
            self._k('aln_sem_heads_bwd', C.byref(P.descs['semf']), C.byref(P.descs['semo']), H.ptr(c['sigma_out']), H.ptr(c['feat']),
                    H.ptr(c['w_row']), H.ptr(g_sem), H.ptr(g_feat), N, S1, S2, L.C, M, L.D, L.G, H.ptr(d_oin), H.ptr(d_fin),
                    gp('semf'), gp('semo'), 1, None, fi, H.stream(), tag=('sem', M))
            d_oin = None     # fold_geo = 1: its geo_feat columns are already inside d_fin
        elif lin:
            # semantic_features layers 2 and 1 from d h2 (the last layer was handled per ray above; semantic_out took no part)
            fs = nets['semf']
            Wft = P.wide_wt['semf']
            h1, h2, _ = c['wide_saved']
            of = L.offsets['semf']
            n0 = fs.shapes[0][0] * fs.shapes[0][1]
            dh1 = g('wide_dh1', (M, fs.hidden), f16)
            if h1 is None:
                self._nt_maskgen(M, Wft[1], dh1, d_feat, c['sigma_out'], tag=('sem', M))
                self._tn_gen(M, d_feat, of + n0, c['sigma_out'], tag=('sem', M))
            else:
                self._nt(M, fs.hidden, Wft[1], dh1, a1=d_feat, K1=fs.hidden, mask=h1, watch=True, tag=('sem', M))
                self._tn(M, fs.hidden, d_feat, of + n0, fs.hidden, a1=h1, K1=fs.hidden, tag=('sem', M))
            d_fin = g('d_semf_in', (M, fs.in_pad), f16)
            if self._din_fused():
                self._tn_din(M, dh1, d_fin, c['sigma_out'], of, tag=('sem', M))
            else:
                self._nt(M, fs.in_pad, Wft[0], d_fin, a1=dh1, K1=fs.hidden, watch=True, tag=('sem', M))
                self._tn(M, fs.hidden, dh1, of, fs.in_pad, geo=c['sigma_out'], tag=('sem', M))
            d_oin = None
        elif sem and c.get('sem_wide'):
            d_fin, d_oin = self.wide_sem_bwd(c['sigma_out'], M, c['feat'], c['wide_saved'], d_logits, d_feat, lambda n, shp: g(n, shp, f16))
        elif sem:
            fs, os_ = nets['semf'], nets['semo']
            d_oin = g('d_semo_in', (M, os_.in_pad), f16)
            H.call('aln_mlp_bwd', C.byref(P.descs['semo']), H.ptr(c['semo_in']), hp(c['oh1']), None, H.ptr(d_logits), M, None,
                   H.ptr(dA1), None, H.ptr(d_oin), gp('semo'), fi, H.stream())
            H.call('aln_assemble_dsemf_out', H.ptr(d_feat), H.ptr(c['feat']), H.ptr(d_oin), M, L.D, os_.in_pad, fi, H.stream())
            d_fin = g('d_semf_in', (M, fs.in_pad), f16)
            H.call('aln_mlp_bwd', C.byref(P.descs['semf']), H.ptr(c['semf_in']), hp(c['fh1']), hp(c['fh2']), H.ptr(d_feat), M,
                   None, H.ptr(dA1), H.ptr(dA2), H.ptr(d_fin), gp('semf'), fi, H.stream())
        wide = sem and c.get('sem_wide') and not lin   # the wide path hands over the 16 geo columns of d(semo_in) only
        d_enc = g('d_enc', (M, e.enc_pad), f16)
        if c.get('enc_planes'):
            P.desc_sigma_planes.x_pitch = M
        sigma_desc = P.desc_sigma_planes if c.get('enc_planes') else P.desc_sigma_tiled if c.get('enc_tiled') else P.descs['sigma']
        # the density head's dL/dout rows [d_h0 | d(geo_feat) of the semantic pair + of the colour head] are built by its backward kernel's
        # own loader (aln_mlp_bwd_dso) where that kernel exists: no aln_assemble_grads pass, no d_sigma_out buffer
        dso = (self.fold_dsigma and rc and d_fin is not None and d_oin is None and nets['semf'].in_pad == 16 and cs.in_pad == 32 and
               nets['sigma'].in_pad == 48 and nets['sigma'].hidden == 128 and nets['sigma'].n_hidden == 2)
        if dso:
            self._k('aln_mlp_bwd_dso', C.byref(sigma_desc), H.ptr(c['enc']), H.ptr(d_h0), H.ptr(d_fin), H.ptr(d_cin), H.ptr(c['cidx_row']), L.G, M,
                    H.ptr(d_enc), gp('sigma'), fi, H.stream(), tag=('sigma', M))
        else:
            d_sout = g('d_sigma_out', (M, 16), f16)
            H.call('aln_assemble_grads', H.ptr(d_h0), H.ptr(d_fin), nets['semf'].in_pad, H.ptr(d_oin), 16 if wide else nets['semo'].in_pad,
                   0 if wide else L.D, H.ptr(d_cin), cs.in_pad, H.ptr(c['cidx_row']), M, L.G, H.ptr(d_sout), fi, H.stream())
            self._k('aln_mlp_bwd', C.byref(sigma_desc), H.ptr(c['enc']), hp(c['h1']), hp(c['h2']), H.ptr(d_sout), M, None,
                    H.ptr(dA1), H.ptr(dA2), H.ptr(d_enc), gp('sigma'), fi, H.stream(), tag=('sigma', M))
        ro, rd, z = c['rays_o'], c['rays_d'], c['z']
        if rc:   # all fused heads' weight-gradient slabs -> P.grad, one launch (fixed summation order: bit-reproducible)
            heads = [k for k in ('color', 'semf', 'semo', 'sigma') if k in P.descs and (sem or k in ('color', 'sigma'))]
            n = len(heads)
            ds = (C.c_void_p * n)(*[C.addressof(P.descs[k]) for k in heads])
            dws = (C.c_void_p * n)(*[P.grad.data_ptr() + 4 * L.offsets[k] for k in heads])
            rows = (C.c_int32 * n)(*([M] * n))
            # the fused semantic pair (one kernel for both heads) leaves its own number of slabs
            pair = int(H.lib().aln_sem_heads_bwd_slabs(C.byref(P.descs['semf']), C.byref(P.descs['semo']), M, L.D, L.G)) if sem_fused else 0
            slabs = (C.c_int32 * n)(*[pair if k in ('semf', 'semo') else 0 for k in heads])
            self._k('aln_mlp_dw_reduce_slabs', n, ds, dws, rows, slabs, H.stream(), tag=None)
        if on_grad_ready is not None:
            on_grad_ready('mlp', L.n_grid, L.n_total)
        if not e.use_grid:
            return
        # binned scatter (encode.hip): both passes in one launch pair per level group, no global atomics, bit-reproducible
        nl = int(e.grid.n_levels)
        bins = ws.scratch('enc_bwd_bins', H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M))
        F = int(e.grid.n_features)
        flag = H.ptr(scatter_flag) if scatter_flag is not None else fi
        split = level_groups is not None and len(level_groups) > 1 and grid_adam is None
        if split:   # phase 1 ONCE for every level of the groups (one launch instead of one per group), phase 2 group by group below
            self._k('aln_encode_bwd_binned_phase', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, max(S2, 1),
                    H.ptr(c['perm']) if S2 > 0 else None, H.ptr(d_enc), None, H.ptr(bins), min(lo for lo, _ in level_groups), max(hi for _, hi in level_groups),
                    flag, None, 0.0, 1, H.stream(), tag=(M, nl))
        for lo, hi in (level_groups or [(0, nl)]):
            # (two passes: the tiles walk every ray in depth order, so coarse and fine samples of one cell dedupe into one record)
            if split:
                self._k('aln_encode_bwd_binned_phase', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, max(S2, 1),
                        H.ptr(c['perm']) if S2 > 0 else None, H.ptr(d_enc), H.ptr(P.grad) if grid_wire is None else None, H.ptr(bins), lo, hi, flag,
                        H.ptr(grid_wire), float(wire_mul), 2, H.stream(), tag=(M, hi - lo))
            elif grid_wire is not None:
                assert grid_adam is None and grid_wire.dtype == torch.float16 and grid_wire.numel() >= L.n_grid
                self._k('aln_encode_bwd_binned_wire', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, max(S2, 1),
                        H.ptr(c['perm']) if S2 > 0 else None, H.ptr(d_enc), H.ptr(bins), lo, hi, H.ptr(scatter_flag) if scatter_flag is not None else fi,
                        H.ptr(grid_wire), float(wire_mul), H.stream(), tag=(M, hi - lo))
            else:
                self._k('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, max(S2, 1),
                        H.ptr(c['perm']) if S2 > 0 else None, H.ptr(d_enc), H.ptr(P.grad), H.ptr(bins), lo, hi, H.ptr(scatter_flag) if scatter_flag is not None else fi,
                        C.byref(grid_adam) if grid_adam is not None else None, H.stream(), tag=(M, hi - lo))
            if on_grad_ready is not None and level_groups is not None:
                a = int(e.grid.offset[lo]) * F
                b = int(e.grid.offset[hi]) * F if hi < nl else L.n_grid
                on_grad_ready('grid', a, b)
