"""Seeded synthetic RGB-D scenes in the reference's scene layout.

S0  ``make_cube_scene``  32x32, 12 frames on a ring around a coloured / labelled unit cube.
S1  ``make_room_scene``  640x480 Replica-style pinhole (fx=fy=320, cx=319.5, cy=239.5, scripts/convert_replica.py:
    106-116), smooth random walk inside a 6x4x3 m textured box room holding 5 labelled boxes, depth in
    uint16 millimetres with 2 % zero holes, optional DINO-/LSeg-like feature maps [F,60,80,Cf] f16.

Arrays follow autolabel/dataset.py:352-418: images [F,H*W,3] f32 in [0,1], depths [F,H*W] u16 mm, semantics
[F,H*W] u8 (0 = unlabeled, 1 = background, 2.. = objects), poses T_CW (world -> OpenCV camera) 4x4.
Everything is computed with torch on the given device so a bench can build it directly in HBM.
"""
import math

import numpy as np
import torch


def _look_at(eye, target, up):
    """T_CW for an OpenCV camera (x right, y down, z forward)."""
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(z, up)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R_WC = torch.stack([x, y, z], 1)  # columns = camera axes in world
    T_WC = torch.eye(4, dtype=torch.float64)
    T_WC[:3, :3] = R_WC
    T_WC[:3, 3] = eye
    return torch.linalg.inv(T_WC)


def _slab(o, d, lo, hi):
    inv = 1.0 / d
    t1, t2 = (lo - o) * inv, (hi - o) * inv
    tn = torch.minimum(t1, t2).max(-1).values
    tf = torch.maximum(t1, t2).min(-1).values
    return tn, tf


def _texture(p, phase):
    """Smooth procedural colour in [0,1] from world position."""
    f = torch.tensor([[2.1, 3.3, 1.7], [1.3, 2.7, 3.9], [3.1, 1.1, 2.3]], dtype=p.dtype, device=p.device)
    return 0.5 + 0.45 * torch.sin(p @ f.t() + phase)


def _render_frames(T_CWs, w, h, fx, fy, cx, cy, room, boxes, device, hole_frac, gen):
    F = len(T_CWs)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing='ij')
    d_cam = torch.stack([(xs + 0.5 - cx) / fx, (ys + 0.5 - cy) / fy, torch.ones_like(xs)], -1).reshape(-1, 3).to(device)
    zscale = 1.0 / d_cam.norm(dim=1)  # z-depth = t * zscale for unit directions
    images = torch.empty(F, h * w, 3, dtype=torch.float32, device=device)
    depths = torch.empty(F, h * w, dtype=torch.int32, device=device)
    sems = torch.empty(F, h * w, dtype=torch.uint8, device=device)
    lo, hi = [torch.tensor(v, dtype=torch.float64, device=device) for v in room]
    for f, T_CW in enumerate(T_CWs):
        T_WC = torch.linalg.inv(T_CW).to(device)
        o = T_WC[:3, 3]
        d = d_cam @ T_WC[:3, :3].t()
        d = d / d.norm(dim=1, keepdim=True)
        _, t_hit = _slab(o, d, lo, hi)  # camera is inside the room: exit point
        label = torch.ones(h * w, dtype=torch.uint8, device=device)
        phase = torch.zeros(h * w, dtype=torch.float64, device=device)
        for bi, (blo, bhi, ph) in enumerate(boxes):
            tn, tf = _slab(o, d, blo.to(device), bhi.to(device))
            hit = (tn < tf) & (tn > 0) & (tn < t_hit)
            t_hit = torch.where(hit, tn, t_hit)
            label = torch.where(hit, torch.tensor(2 + bi, dtype=torch.uint8, device=device), label)
            phase = torch.where(hit, torch.tensor(ph, dtype=torch.float64, device=device), phase)
        p = o + d * t_hit[:, None]
        images[f] = _texture(p, phase[:, None]).float()
        depths[f] = (t_hit * zscale * 1000.0).round().clamp(0, 65535).int()
        sems[f] = label
    if hole_frac > 0:
        holes = torch.rand(depths.shape, generator=gen, device='cpu') < hole_frac
        depths[holes.to(device)] = 0
    return images, depths, sems


def make_room_scene(n_frames=200, w=640, h=480, fx=320.0, fy=320.0, cx=319.5, cy=239.5, seed=0, device='cpu',
                    labelled_every=10, feat_dim=0, feat_hw=(60, 80), hole_frac=0.02):
    gen = torch.Generator().manual_seed(seed)
    room = ([-3.0, -2.0, -1.5], [3.0, 2.0, 1.5])
    boxes = []
    for i in range(5):
        c = torch.tensor([-2.2 + 1.1 * i, (-1.0) ** i * 0.9, -1.5 + 0.4], dtype=torch.float64)
        half = torch.tensor([0.35, 0.3, 0.4], dtype=torch.float64)
        boxes.append((c - half, c + half, 1.0 + 1.3 * i))
    # smooth random walk of eye and gaze target
    eye = torch.zeros(3, dtype=torch.float64)
    tgt = torch.tensor([2.0, 0.0, -0.5], dtype=torch.float64)
    ve, vt = torch.zeros(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
    lim_e = torch.tensor([2.0, 1.2, 0.6], dtype=torch.float64)
    lim_t = torch.tensor([2.8, 1.8, 1.2], dtype=torch.float64)
    T_CWs = []
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    for _ in range(n_frames):
        ve = 0.9 * ve + 0.06 * torch.randn(3, generator=gen, dtype=torch.float64)
        vt = 0.9 * vt + 0.15 * torch.randn(3, generator=gen, dtype=torch.float64)
        eye = torch.clamp(eye + ve, -lim_e, lim_e)
        tgt = torch.clamp(tgt + vt, -lim_t, lim_t)
        if (tgt - eye).norm() < 0.5:
            tgt = tgt + torch.tensor([1.0, 0.3, 0.0], dtype=torch.float64)
        T_CWs.append(_look_at(eye, tgt, up))
    images, depths, sems_full = _render_frames(T_CWs, w, h, fx, fy, cx, cy, room, boxes, device, hole_frac, gen)
    sems = torch.zeros_like(sems_full)
    sems[::labelled_every] = sems_full[::labelled_every]
    scene = dict(images=images, depths=depths, semantics=sems, semantics_full=sems_full,
                 T_CW=torch.stack(T_CWs).numpy(), w=w, h=h, intrinsics=(fx, fy, cx, cy), n_classes=7,
                 min_bounds=np.array(room[0]), max_bounds=np.array(room[1]))
    if feat_dim:
        Hf, Wf = feat_hw
        ys = ((torch.arange(Hf) + 0.5) * h / Hf).long().clamp(max=h - 1)
        xs = ((torch.arange(Wf) + 0.5) * w / Wf).long().clamp(max=w - 1)
        idx = (ys[:, None] * w + xs[None]).reshape(-1).to(device)
        onehot = torch.nn.functional.one_hot(sems_full[:, idx].long(), 7).float()
        base = torch.cat([images[:, idx], onehot], -1)  # [F, Hf*Wf, 10]
        proj = torch.randn(10, feat_dim, generator=gen).to(device)
        feats = base @ proj
        if feat_dim >= 256:  # LSeg-like: unit norm (scripts/ros/node.py:105)
            feats = feats / feats.norm(dim=-1, keepdim=True)
        scene.update(features=feats.half(), feat_hw=(Hf, Wf))
    return scene


def make_cube_scene(n_frames=12, size=32, seed=0, device='cpu'):
    """S0: unit cube, 6 distinctly coloured / labelled faces, ring of cameras at radius 2.5."""
    gen = torch.Generator().manual_seed(seed)
    f = 16.0 / math.tan(math.radians(45.0))
    T_CWs = []
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    for i in range(n_frames):
        a = 2 * math.pi * i / n_frames
        eye = torch.tensor([2.5 * math.cos(a), 2.5 * math.sin(a), 0.8], dtype=torch.float64)
        T_CWs.append(_look_at(eye, torch.zeros(3, dtype=torch.float64), up))
    room = ([-3.0, -3.0, -3.0], [3.0, 3.0, 3.0])  # cameras (radius 2.5) sit inside the room; bbox = room
    half = torch.tensor([0.5, 0.5, 0.5], dtype=torch.float64)
    boxes = [(-half, half, 2.0)]
    images, depths, sems_full = _render_frames(T_CWs, size, size, f, f, size / 2 - 0.5, size / 2 - 0.5, room, boxes, device, 0.0, gen)
    sems = torch.zeros_like(sems_full)
    sems[:2] = sems_full[:2]
    return dict(images=images, depths=depths, semantics=sems, semantics_full=sems_full, T_CW=torch.stack(T_CWs).numpy(),
                w=size, h=size, intrinsics=(f, f, size / 2 - 0.5, size / 2 - 0.5), n_classes=3,
                min_bounds=np.array(room[0]), max_bounds=np.array(room[1]))


def subsample(scene, factor):
    """Nearest-neighbour resize by an integer factor (cv2.INTER_NEAREST of dataset.py:368-370) + scaled intrinsics
    (autolabel/utils/__init__.py:13-20)."""
    factor = int(factor)
    if factor == 1:
        return scene
    w, h = scene['w'], scene['h']
    nw, nh = w // factor, h // factor
    ys = (torch.arange(nh) * factor).long()
    xs = (torch.arange(nw) * factor).long()
    idx = (ys[:, None] * w + xs[None]).reshape(-1).to(scene['images'].device)
    fx, fy, cx, cy = scene['intrinsics']
    sx, sy = nw / w, nh / h
    out = dict(scene)
    out.update(images=scene['images'][:, idx].contiguous(), depths=scene['depths'][:, idx].contiguous(),
               semantics=scene['semantics'][:, idx].contiguous(), w=nw, h=nh,
               intrinsics=(fx * sx, fy * sy, cx * sx, cy * sy))
    if scene.get('semantics_full') is not None:   # dense ground truth of the unlabelled frames (evaluation only)
        out['semantics_full'] = scene['semantics_full'][:, idx].contiguous()
    return out
