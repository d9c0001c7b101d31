"""Size-independent properties of the hot path at BASELINE.json's full sizes (B = 8192 rays x 256 samples, hash grid
L = 16, T = 2^19, 640x480 frames): the oracle needs minutes for these shapes, so here the kernels are checked against
identities that hold for any input -- grid-vertex lookups, forward/backward adjointness, additivity over rays, ray
independence, unit norms -- next to the oracle comparisons at small sizes in test_gpu_kernels.py / test_gpu_pipeline.py.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as O

B_FULL, S_FULL = 8192, 256          # BASELINE configs[0] batch; 128 + 128 samples per ray


@pytest.fixture(scope='module')
def H():
    from autolabel_amd import hip
    hip.require_gpu()
    hip.lib()
    return hip


def _rays(n, bound, seed):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(n, 3, generator=g) - 0.5) * bound
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    return o.cuda(), d.cuda()


def test_grid_vertices_return_table_entries_bit_exact(H):
    """At a vertex of level l the trilinear weights are (1, 0, ..., 0): the feature IS the table entry the index function
    (dense stride or the xor-of-primes hash) selects -- 2^20 lookups per level against the oracle's index function."""
    bound = 3.0
    e = H.make_enc_desc('hg+freq', bound)
    spec = O.GridSpec()
    g = torch.Generator().manual_seed(0)
    table = (torch.rand(spec.n_entries, 2, generator=g) - 0.5).half()
    td = table.cuda()
    n = 1 << 20
    for lvl in (0, 2, 3, 9, 15):          # dense, last dense, first hashed, mid, finest
        L = spec.levels()[lvl]
        gi = torch.randint(0, int(L['res']) - 1, (n, 3), generator=g)
        # pos = xn * scale + 0.5 must be the integer gi: xn = (gi - 0.5) / scale, kept where that round trip is exact in fp32
        xn = (gi.float() - 0.5) / np.float32(L['scale'])
        ok = ((xn * np.float32(L['scale']) + np.float32(0.5)) == gi.float()).all(dim=1) & (xn >= 0).all(dim=1)
        x = (xn * np.float32(2 * bound) - np.float32(bound))
        ok &= (((x + np.float32(bound)) / np.float32(2 * bound)) == xn).all(dim=1)
        assert ok.float().mean() > 0.2, lvl
        x, gi = x[ok].contiguous(), gi[ok]
        out = torch.zeros(x.shape[0], e.enc_pad, dtype=torch.float16, device='cuda')
        xd = x.cuda()
        H.call('aln_encode_fwd', C.byref(e), H.ptr(td), None, None, None, H.ptr(xd), x.shape[0], 1, H.ptr(out), H.stream())
        idx, wgt = O.grid_corner_indices(xn[ok].contiguous(), L)   # oracle index function; corner 0 of a vertex = the vertex
        assert bool((wgt[:, 0] == 1).all()) and bool((wgt[:, 1:] == 0).all())
        want = table[int(L['offset']) + idx[:, 0]]
        got = out.cpu()[:, 12 + 2 * lvl:12 + 2 * lvl + 2]
        assert torch.equal(got, want), f'level {lvl}: vertex lookups must be bit-exact'


def test_binned_scatter_is_the_adjoint_of_the_gather_at_full_size(H):
    """<encode(T), D> == <T, encode_bwd(D)> for the grid part (the scatter is the transpose of the gather), both renderer passes
    in one launch: 8192 rays x (128 + 128) samples = 2^21 rows.  Record values are fp16 (2^-11 each), their sums are exact."""
    bound = 3.0
    e = H.make_enc_desc('hg+freq', bound)
    spec = O.GridSpec()
    N, S1, S2 = B_FULL, S_FULL // 2, S_FULL // 2
    M1, M = N * S1, N * (S1 + S2)
    g = torch.Generator().manual_seed(7)
    ro, rd = _rays(N, bound, 4)
    z = torch.cat([(torch.rand(N, S1, generator=g).sort(dim=1)[0] * 5 + 0.2).reshape(-1),
                   (torch.rand(N, S2, generator=g).sort(dim=1)[0] * 5 + 0.2).reshape(-1)]).cuda().contiguous()
    table = ((torch.rand(spec.n_entries, 2, generator=g) - 0.5)).half().cuda()
    enc = torch.zeros(M, e.enc_pad, dtype=torch.float16, device='cuda')
    H.call('aln_encode_fwd', C.byref(e), H.ptr(table), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M1, S1, H.ptr(enc), H.stream())
    H.call('aln_encode_fwd', C.byref(e), H.ptr(table), H.ptr(ro), H.ptr(rd), H.ptr(z[M1:]), None, M - M1, S2, H.ptr(enc[M1:]), H.stream())
    d_enc = torch.zeros(M, e.enc_pad, dtype=torch.float16, device='cuda')
    d_enc[:, 12:44] = (torch.randn(M, 32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) * 0.05).half()
    ws = torch.empty(int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M)), dtype=torch.uint8, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')

    def binned(grad, a=0, b=N, lo=0, hi=16):
        # rays [a, b): their coarse rows followed by their fine rows, as one launch
        zz = torch.cat([z[a * S1:b * S1], z[M1 + a * S2:M1 + b * S2]]).contiguous()
        de = torch.cat([d_enc[a * S1:b * S1], d_enc[M1 + a * S2:M1 + b * S2]]).contiguous()
        n = b - a
        H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro[a:b]), H.ptr(rd[a:b]), H.ptr(zz), None, n * (S1 + S2), n * S1, S1, S2,
               None, H.ptr(de), H.ptr(grad), H.ptr(ws), lo, hi, H.ptr(flag), None, H.stream())

    grad = torch.zeros(spec.n_entries * 2, device='cuda')
    binned(grad)
    lhs = (enc[:, 12:44].double() * d_enc[:, 12:44].double()).sum().item()
    rhs = (table.double().reshape(-1) * grad.double()).sum().item()
    scale = (enc[:, 12:44].double().abs() * d_enc[:, 12:44].double().abs()).sum().item()
    assert abs(lhs - rhs) <= 2.0 ** -10 * scale * 0.05 + 1e-6 * scale, (lhs, rhs, scale)
    assert flag.item() == 0
    # bit-reproducible at full size: every level, every entry (integer accumulation, one owner block per entry)
    again = torch.zeros_like(grad)
    binned(again)
    assert torch.equal(again, grad)
    # additivity over rays (the per-tile scaling of the records changes with the tiling: fp16 record rounding only)
    g2 = torch.zeros_like(grad)
    binned(g2, 0, N // 2); binned(g2, N // 2, N)
    assert (g2 - grad).abs().max().item() <= 1e-4 * max(1.0, grad.abs().max().item())
    # level groups (the data-parallel launch order) tile the scatter exactly
    g3 = torch.zeros_like(grad)
    for lo, hi in ((12, 16), (8, 12), (4, 8), (0, 4)):
        binned(g3, lo=lo, hi=hi)
    assert torch.equal(g3, grad)
    # the depth-order walk of the training step (coarse and fine samples of a ray interleaved by depth): the adjoint identity and
    # the reproducibility hold for it too, with fewer records than the pass-major walk
    order = torch.cat([z[:M1].view(N, S1), z[M1:].view(N, S2)], 1).argsort(dim=1, stable=True).to(torch.int16).contiguous()

    def records():
        tile = int(H.lib().aln_encode_bwd_binned_tile_rows())
        nt, nl = (M + tile - 1) // tile, 16
        pool = int(H.lib().aln_encode_bwd_binned_ws_bytes(C.byref(e), M)) - nl * 64 * nt * 4    # (the descriptors close the workspace)
        return int(((ws[pool:pool + nl * 64 * nt * 4].view(torch.int32) >> 13) & 0x3FFF).sum().item())
    n_plain = records()     # (of the last pass-major launch over all levels... the level groups wrote the same descriptors)
    walk, walk2 = torch.zeros_like(grad), torch.zeros_like(grad)
    for dst in (walk, walk2):
        H.call('aln_encode_bwd_binned', C.byref(e), H.ptr(ro), H.ptr(rd), H.ptr(z), None, M, M1, S1, S2, H.ptr(order), H.ptr(d_enc), H.ptr(dst),
               H.ptr(ws), 0, 16, H.ptr(flag), None, H.stream())
    assert torch.equal(walk, walk2) and flag.item() == 0
    rhs_w = (table.double().reshape(-1) * walk.double()).sum().item()
    assert abs(lhs - rhs_w) <= 2.0 ** -10 * scale * 0.05 + 1e-6 * scale, (lhs, rhs_w, scale)
    assert (walk - grad).norm().item() <= 1e-3 * grad.norm().item()
    assert records() < n_plain


def test_render_is_ray_independent_and_bounded_at_full_batch():
    """B = 8192 rays, 128 + 128 samples: permuting the rays permutes the outputs (no cross-ray coupling anywhere in the
    launch sequence), the weights of a ray sum to at most one, and depth lies inside [near, far] / norm."""
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, 7, bound=3.0)
    P = Params(layout, 'cuda'); P.init_(seed=0)
    with torch.no_grad():
        P.flat[:layout.n_grid].mul_(3e3)
    P.refresh_shadows()
    pipe = HipPipeline(layout, P)
    N = B_FULL
    ro, rd = _rays(N, 3.0, 5)
    norms = (torch.rand(N, device='cuda') * 0.4 + 1.0)
    out, c = pipe.forward(ro, rd, norms, 128, 128, False, train=False)
    base = {k: v.clone() for k, v in out.items()}
    near, far = c['nears'].clone(), c['fars'].clone()
    assert torch.isfinite(torch.cat([v.reshape(-1) for v in base.values()])).all()
    assert base['weights_sum'].max().item() <= 1.0 + 1e-5 and base['weights_sum'].min().item() >= 0.0
    hit = base['weights_sum'] > 0.5
    d = base['depth'] * norms
    assert (d[hit] <= far[hit] + 1e-4).all() and (d[hit] >= near[hit] * base['weights_sum'][hit] - 1e-4).all()
    perm = torch.randperm(N, device='cuda', generator=torch.Generator(device='cuda').manual_seed(9))
    out2, _ = pipe.forward(ro[perm].contiguous(), rd[perm].contiguous(), norms[perm].contiguous(), 128, 128, False, train=False)
    for k in ('image', 'depth', 'semantic', 'semantic_features', 'weights_sum', 'coordinates_map', 'depth_variance'):
        assert torch.equal(out2[k], base[k][perm]), f'{k}: a ray must not depend on its position in the batch'


def test_full_frame_rays_are_unit_length_and_norms_follow_the_pinhole_formula(H):
    """640x480 frame (307200 rays): |d| == 1 to one fp32 ulp, norm == ||((x+.5-cx)/fx, (y+.5-cy)/fy, 1)||, and the sum
    of all directions equals R_WC applied to the sum of the normalised camera-frame rays (linearity of the rotation)."""
    w, h, fx, fy, cx, cy = 640, 480, 320.0, 320.0, 319.5, 239.5
    g = torch.Generator().manual_seed(4)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    R = q.float().contiguous()
    idx = torch.arange(w * h, dtype=torch.int64)
    dirs = torch.empty(w * h, 3, device='cuda'); norms = torch.empty(w * h, device='cuda')
    Rd, idxd = R.cuda(), idx.cuda()
    H.call('aln_compute_direction', H.ptr(Rd), H.ptr(idxd), w * h, w, fx, fy, cx, cy, None, H.ptr(dirs), H.ptr(norms), H.stream())
    x = (idx % w).double() + 0.5; y = (idx // w).double() + 0.5
    cam = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], dim=1)
    n64 = cam.norm(dim=1)
    assert (norms.cpu().double() - n64).abs().max().item() <= 2e-7 * n64.max().item()
    assert (dirs.cpu().double().norm(dim=1) - 1).abs().max().item() <= 3e-7
    want_sum = (R.double() @ (cam / n64[:, None]).sum(dim=0))
    assert (dirs.cpu().double().sum(dim=0) - want_sum).abs().max().item() <= 1e-2   # 307200 fp32 roundings


@pytest.mark.parametrize('encoding', ['hg+freq', 'hg'])
def test_level_phased_forward_encode_is_bit_identical_to_the_single_kernel(H, encoding):
    """The level-phased forward (two kernels, per-level planes) must reproduce the tile kernel bit for bit, including a
    ragged tail (rows not a multiple of 256) and point (xyz) input."""
    bound = 3.0
    e = H.make_enc_desc(encoding, bound)
    spec = O.GridSpec()
    g = torch.Generator().manual_seed(6)
    table = ((torch.rand(spec.n_entries, 2, generator=g) - 0.5)).half().cuda()
    for N, S, use_xyz in ((2048, 128, False), (777, 93, False), (70001, 1, True)):
        rows = N * S
        ro, rd = _rays(N, bound, 7)
        z = (torch.rand(N, S, generator=g).sort(dim=1)[0] * 5 + 0.2).cuda().reshape(-1).contiguous()
        xyz = ((torch.rand(rows, 3, generator=g) * 2 - 1) * bound * 1.05).cuda() if use_xyz else None
        a = torch.zeros(rows, e.enc_pad, dtype=torch.float16, device='cuda')
        b = torch.zeros_like(a)
        ws = torch.empty(H.lib().aln_encode_fwd_ws_bytes(C.byref(e), rows), dtype=torch.uint8, device='cuda')
        args = (H.ptr(table), None, None, None, H.ptr(xyz)) if use_xyz else (H.ptr(table), H.ptr(ro), H.ptr(rd), H.ptr(z), None)
        H.call('aln_encode_fwd', C.byref(e), *args, rows, S, H.ptr(a), H.stream())
        H.call('aln_encode_fwd_phased', C.byref(e), *args, rows, S, H.ptr(ws), H.ptr(b), H.stream())
        assert torch.equal(a, b), (encoding, N, S)
