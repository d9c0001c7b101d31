"""CPU checks of the occupancy-marching oracle (oracle/march_oracle.py) and of the quality metrics the bench reports."""
import numpy as np
import torch

from oracle import march_oracle as MO


def _case(N=60, G=16, fill=0.3, seed=0):
    g = np.random.default_rng(seed)
    bound = 2.0
    o = ((g.random((N, 3)) - 0.5) * 3).astype(np.float32)
    d = g.normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[::9] += 50.0          # rays that miss the box
    return bound, o, d, g.random(G ** 3) < fill


def test_march_rows_cover_the_occupied_steps():
    bound, o, d, bits = _case()
    S, ms, G = 12, 128, 16
    near, far, z, delta, counts = MO.march_rays(o, d, S, bound, 0.2, bits, G, ms)
    dt = np.float32(3.4641016151377544) * np.float32(bound) / np.float32(ms)
    assert (np.diff(z, axis=1) >= 0).all(), 'rows are ordered along the ray'
    for r in range(len(o)):
        K = counts[r]
        live = delta[r] > 0
        if K == 0:
            assert not live.any() and (z[r] == near[r]).all()
        elif K <= S:
            assert live.sum() == K and np.allclose(delta[r][live], dt) and (z[r, K:] == z[r, K - 1]).all()
        else:   # subsampled: all S rows live, the step lengths add up to the occupied length K * dt
            assert live.all() and np.isclose(delta[r].sum(), K * dt, rtol=1e-5)
        # every live row sits in an occupied cell
        p = o[r][None] + d[r][None] * z[r][live, None]
        assert bits[MO.cell_of(p, bound, G)].all()
    assert (counts == 0).any() and (counts > S).any()
    # packed-word input gives the same rows
    words = (np.pad(bits, (0, (-len(bits)) % 32)).reshape(-1, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(1).astype(np.uint32)
    z2 = MO.march_rays(o, d, S, bound, 0.2, words, G, ms)[2]
    assert np.array_equal(z, z2)


def test_grid_update_and_untrained_cells():
    G, bound = 8, 1.0
    g = np.random.default_rng(1)
    grid = g.random(G ** 3).astype(np.float32)
    grid[::5] = -1
    sigma = (g.random(G ** 3) * 3).astype(np.float32)
    new, bits, mean = MO.grid_update(grid, sigma, 0.95, 1.0, 10.0)
    live = grid >= 0
    assert (new[~live] == -1).all() and not bits[~live].any()
    assert np.allclose(new[live], np.maximum(grid[live] * np.float32(0.95), sigma[live]))
    assert np.array_equal(bits[live], new[live] > mean)            # thresh 10 > mean: the mean decides
    pts = MO.grid_points(G, bound, g.random((G ** 3, 3)).astype(np.float32) * 0.999)
    assert np.array_equal(MO.cell_of(pts, bound, G), np.arange(G ** 3))
    T = np.eye(4, dtype=np.float32)[None]                        # one camera at the origin looking down +z
    m = MO.mark_untrained(np.zeros(G ** 3, np.float32), G, bound, T, 4.0, 4.0, 3.5, 3.5, 8.0, 8.0, 0.0, 2)
    cz = (np.arange(G ** 3) // (G * G))
    assert (m[cz < G // 2 - 1] == -1).all() and (m[cz >= G // 2] >= 0).any()     # nothing behind the camera is trainable


def test_heldout_metrics_definition():
    """PSNR / depth L1 / mIoU of autolabel_amd.quality on a hand-made frame (mIoU = mean over classes of intersection / union
    of the argmax, autolabel/evaluation.py:21-29)."""
    from autolabel_amd.quality import heldout_metrics

    class Frames:
        n_frames = 1

        def get_test(self, i):
            return {'rays_o': torch.zeros(4, 3), 'rays_d': torch.zeros(4, 3), 'direction_norms': torch.ones(4),
                    'pixels': torch.tensor([[0.5, 0.5, 0.5]] * 4), 'depth': torch.tensor([1.0, 2.0, 0.0, 1.0]),
                    'semantic': torch.tensor([0, 1, 1, -1])}

    def render(ro, rd, dn):
        logits = torch.tensor([[2.0, 0.0], [0.0, 3.0], [1.0, 0.0], [0.0, 1.0]])    # argmax: 0, 1, 0, 1
        return {'image': torch.tensor([[0.5, 0.5, 0.5]] * 3 + [[0.6, 0.5, 0.5]]), 'depth': torch.tensor([1.5, 2.0, 9.0, 1.0]), 'semantic': logits}
    q = heldout_metrics(render, Frames(), 2)
    assert abs(q['psnr_db'] - (-10 * np.log10(0.01 / 12))) < 1e-4
    assert abs(q['depth_l1_m'] - 0.5 / 3) < 1e-6                    # depth 0 is a hole: not scored
    # labelled pixels 0,1,2: class 0: inter 1 (pixel 0), union 2 (pixels 0, 2) ; class 1: inter 1, union 2
    assert abs(q['miou'] - 0.5) < 1e-9 and q['classes_scored'] == 2
