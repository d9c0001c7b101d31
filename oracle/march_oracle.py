"""CPU restatement (numpy, fp32 op by op) of the occupancy-grid marching spec of autolabel_amd/csrc/march.hip.

TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else; the product path never touches it.

Reference anchor: the `cuda_ray` hooks of autolabel/trainer.py:21-23,34-36,176 (`mark_untrained_grid`,
`update_extra_state` every 16 steps) and `NeRFRenderer(cuda_ray=...)` of the torch-ngp fork.  The reference always passes
cuda_ray=False (autolabel/model_utils.py:72) and the fork's source is not under /root/reference, so this is the build's own
spec, modelled on upstream torch-ngp `raymarching` (uniform steps of 2 sqrt(3) bound / max_steps, density grid with decay and
`min(mean, thresh)` occupancy threshold, `-1` for cells no camera sees).  Parity unpinned (nothing to pin against); the HIP
kernels are held bit-exact to THIS file for all index work and step positions.
"""
import numpy as np

f32 = np.float32


def near_far(o, d, bound, min_near):
    """Slab test against [-bound, bound]^3; a miss gives near = far = min_near (same as nerf_oracle.near_far)."""
    o, d = o.astype(f32), d.astype(f32)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = f32(1.0) / d
        t1, t2 = (f32(-bound) - o) * inv, (f32(bound) - o) * inv
    tn = np.fmax.reduce(np.fmin(t1, t2), axis=-1, initial=-np.inf)
    tf = np.fmin.reduce(np.fmax(t1, t2), axis=-1, initial=np.inf)
    miss = ~(tn <= tf)
    near = np.where(miss, f32(min_near), np.maximum(tn, f32(min_near))).astype(f32)
    far = np.maximum(np.where(miss, f32(min_near), tf), near).astype(f32)
    return near, far


def cell_of(p, bound, G):
    """Clamped position [...,3] -> linear cell index ix + G (iy + G iz)."""
    p = np.clip(p.astype(f32), f32(-bound), f32(bound))
    u = ((p + f32(bound)) / (f32(2.0) * f32(bound))) * f32(G)
    i = np.clip(np.floor(u).astype(np.int64), 0, G - 1)
    return i[..., 0] + G * (i[..., 1] + G * i[..., 2])


def march_rays(rays_o, rays_d, S, bound, min_near, bits, G, max_steps, u=None):
    """-> nears[N], fars[N], z[N,S], delta[N,S], counts[N].  bits: bool[G^3] (or packed uint32 words).  u[N]: per-ray jitter in
    [0,1) (perturb) or None = 0.5."""
    rays_o, rays_d = rays_o.astype(f32), rays_d.astype(f32)
    if bits.dtype != np.bool_:
        bits = ((bits.view(np.uint32)[:, None] >> np.arange(32, dtype=np.uint32)[None]) & 1).astype(bool).reshape(-1)[:G ** 3]
    N = rays_o.shape[0]
    near, far = near_far(rays_o, rays_d, bound, min_near)
    dt = f32(3.4641016151377544) * f32(bound) / f32(max_steps)
    z, delta, counts = np.zeros((N, S), f32), np.zeros((N, S), f32), np.zeros(N, np.int32)
    for r in range(N):
        uu = f32(0.5) if u is None else f32(u[r])
        n_steps = int(np.clip(np.ceil((far[r] - near[r]) / dt), 0, max_steps))
        i = np.arange(n_steps, dtype=f32)
        t = near[r] + (i + uu) * dt
        p = rays_o[r][None] + rays_d[r][None] * t[:, None]
        occ = bits[cell_of(p, bound, G)] if n_steps else np.zeros(0, bool)
        tk = t[occ]
        K = len(tk)
        counts[r] = K
        if K <= S:
            z[r, :K], delta[r, :K] = tk, dt
            z[r, K:] = tk[-1] if K else near[r]
        else:
            ranks = np.arange(K, dtype=np.int64)
            j = ranks * S // K
            first = np.ones(K, bool)
            first[1:] = j[1:] != j[:-1]
            z[r, j[first]] = tk[first]
            delta[r, :] = dt * f32(K) / f32(S)
    return near, far, z, delta, counts


def grid_points(G, bound, u):
    """u[G^3,3] uniforms -> jittered cell points [G^3,3]."""
    c = np.arange(G ** 3)
    ci = np.stack([c % G, (c // G) % G, c // (G * G)], 1).astype(f32)
    return ((ci + u.astype(f32)) / f32(G)) * (f32(2.0) * f32(bound)) - f32(bound)


def grid_update(grid, sigma, decay, density_scale, thresh):
    """-> new grid, bool bits, mean.  Cells < 0 (never seen) are left alone and never occupied."""
    grid = grid.astype(f32).copy()
    live = grid >= 0
    if sigma is not None:
        grid[live] = np.maximum(grid[live] * f32(decay), sigma.astype(f32)[live] * f32(density_scale))
    mean = float(grid[live].astype(np.float64).mean()) if live.any() else 0.0
    return grid, grid > min(mean, thresh), mean


def mark_untrained(grid, G, bound, T_CW, fx, fy, cx, cy, w, h, z_near=0.0, sub=2):
    """Cells none of whose sub^3 sub-points falls into some camera's image (in front of it) become -1."""
    grid = grid.astype(f32).copy()
    c = np.arange(G ** 3)
    ci = np.stack([c % G, (c // G) % G, c // (G * G)], 1).astype(f32)
    seen = np.zeros(G ** 3, bool)
    for s in range(sub ** 3):
        si = np.array([s % sub, (s // sub) % sub, s // (sub * sub)], f32)
        p = (ci + (si + f32(0.5)) / f32(sub)) / f32(G) * f32(2.0) * f32(bound) - f32(bound)
        for T in np.asarray(T_CW, f32):
            pc = p @ T[:3, :3].T + T[:3, 3]
            zc = pc[:, 2]
            ok = zc > z_near
            with np.errstate(divide='ignore', invalid='ignore'):
                px, py = fx * pc[:, 0] / zc + cx, fy * pc[:, 1] / zc + cy
            seen |= ok & (px >= 0) & (px <= w) & (py >= 0) & (py <= h)
    grid[~seen] = -1
    grid[seen] = np.maximum(grid[seen], 0)
    return grid
