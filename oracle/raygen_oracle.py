"""numpy restatement of the ray-generation half of autolabel's hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PINNED: every function here is
checked against golden vectors produced by importing the reference's own
``autolabel/dataset.py`` (tests/golden/make_fixtures.py, tests/test_oracle_raygen.py).

    compute_direction   autolabel/dataset.py:17-37
    convert_pose        autolabel/dataset.py:268-274  (+ nerf_matrix_to_ngp, external)
    IndexSampler        autolabel/dataset.py:80-151
    next_train          autolabel/dataset.py:182-242
    get_test            autolabel/dataset.py:244-266
"""
import random

import numpy as np

CV_TO_OPENGL = np.diag([1.0, -1.0, -1.0, 1.0])


def compute_direction(R_WC, ray_indices, w, fx, fy, cx, cy, randomize, jitter=None):
    """dataset.py:17-37.  fx..cy are float64 (np.loadtxt scalars): the pinhole
    division runs in float64 and is rounded once into the float32 output; the
    norm, normalisation and rotation are float32.  ``jitter`` = (jx, jy) float32
    arrays replaces np.random draws when given."""
    n = len(ray_indices)
    directions = np.zeros((n, 3), dtype=np.float32)
    xs = (ray_indices % w).astype(np.float32)
    ys = ((ray_indices - xs) / w).astype(np.float32)
    if randomize:
        if jitter is None:
            jx = np.random.random(n).astype(np.float32)
            jy = np.random.random(n).astype(np.float32)
        else:
            jx, jy = jitter
        xs = xs + jx
        ys = ys + jy
    else:
        xs = xs + np.float32(0.5)
        ys = ys + np.float32(0.5)
    directions[:, 0] = (xs.astype(np.float64) - np.float64(cx)) / np.float64(fx)
    directions[:, 1] = (ys.astype(np.float64) - np.float64(cy)) / np.float64(fy)
    directions[:, 2] = 1.0
    sq = directions * directions
    norm = np.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])[:, None]
    directions = directions / norm
    R = R_WC.astype(np.float32)
    out = np.empty_like(directions)
    for r in range(3):
        out[:, r] = (R[r, 0] * directions[:, 0] + R[r, 1] * directions[:, 1]) + R[r, 2] * directions[:, 2]
    return out, norm


def nerf_matrix_to_ngp(pose, scale=1.0):
    """ashawkey/torch-ngp nerf/provider.py (external): rows (y,z,x), flip cols 1,2."""
    return np.array([
        [pose[1, 0], -pose[1, 1], -pose[1, 2], pose[1, 3] * scale],
        [pose[2, 0], -pose[2, 1], -pose[2, 2], pose[2, 3] * scale],
        [pose[0, 0], -pose[0, 1], -pose[0, 2], pose[0, 3] * scale],
        [0, 0, 0, 1],
    ], dtype=np.float32)


def convert_pose(T_CW):
    """dataset.py:268-274."""
    return nerf_matrix_to_ngp(np.linalg.inv(T_CW) @ CV_TO_OPENGL, scale=1.0)


class IndexSampler:
    """dataset.py:80-151: per-class, per-image pixel index; image prob ~ class pixel count."""

    def __init__(self):
        self.classes = np.array([])
        self.index = {}
        self.image_weights = {}
        self.has_semantics = False
        self.image_range = np.array([])

    def update(self, semantic_maps):
        assert semantic_maps.ndim == 2
        self.index = {}
        classes = np.unique(semantic_maps)
        self.classes = classes[classes != 0]
        counts = {}
        n = len(semantic_maps)
        for i, sem in enumerate(semantic_maps):
            for c in self.classes:
                where = sem == c
                if where.any():
                    self.has_semantics = True
                    self.index.setdefault(c, {})[i] = np.flatnonzero(where)
                    counts.setdefault(c, np.zeros(n))[i] += where.sum()
        self.image_weights = {c: v / v.sum() for c, v in counts.items()}
        self.image_range = np.arange(n, dtype=int)

    def sample_class(self):
        return np.random.choice(self.classes)

    def sample(self, class_id, count=1):
        image_index = np.random.choice(self.image_range, p=self.image_weights[class_id])
        return image_index, np.random.choice(self.index[class_id][image_index], count)

    def semantic_indices(self):
        return sorted({i for d in self.index.values() for i in d})


def next_train(images, depths, semantics, origins, rotations, pixel_indices, sampler, w, intr, batch_size,
               features=None, feat_hw=None, h=None, chunk=512, ratio=0.5):
    """dataset.py:182-242, drawing from the host RNGs in the reference's order."""
    fx, fy, cx, cy = intr
    chunks = batch_size // chunk
    B = chunks * chunk
    out = dict(rays_o=np.zeros((B, 3), np.float32), rays_d=np.zeros((B, 3), np.float32),
               pixels=np.zeros((B, 3), np.float32), direction_norms=np.zeros((B, 1), np.float32),
               depth=np.zeros(B, np.float32), semantic=np.zeros(B, dtype=int))
    if features is not None:
        out['features'] = np.zeros((B, features.shape[-1]), np.float32)
    n_frames = images.shape[0]
    for c in range(chunks):
        if sampler.has_semantics and random.random() < ratio:
            cls = sampler.sample_class()
            fi, idx = sampler.sample(cls, chunk)
        else:
            fi = np.random.randint(0, n_frames)
            idx = np.random.choice(pixel_indices, size=(chunk,))
        s = slice(c * chunk, (c + 1) * chunk)
        out['pixels'][s] = images[fi][idx]
        out['depth'][s] = depths[fi][idx] / 1000.0
        out['semantic'][s] = semantics[fi][idx].astype(int) - 1
        out['rays_o'][s] = origins[fi][None]
        d, nrm = compute_direction(rotations[fi], idx, w, fx, fy, cx, cy, True)
        out['rays_d'][s] = d
        out['direction_norms'][s] = nrm
        if features is not None:
            Hf, Wf = feat_hw
            x = idx % int(w)
            y = (idx - x) / int(w)
            xy = (np.stack([x, y], -1) * np.array([Wf / w, Hf / h])).astype(int)
            out['features'][s] = features[fi][xy[:, 1] * Wf + xy[:, 0], :]
    return out


def get_test(image, depth, semantic, origin, rotation, w, h, intr):
    """dataset.py:244-266 (note: direction_norms stays [H*W,1])."""
    fx, fy, cx, cy = intr
    d, nrm = compute_direction(rotation, np.arange(w * h), w, fx, fy, cx, cy, False)
    return dict(pixels=image.reshape(h, w, 3),
                rays_o=np.broadcast_to(origin, (h, w, 3)).astype(np.float32),
                rays_d=d.reshape(h, w, 3).astype(np.float32),
                depth=(depth / 1000.0).reshape(h, w),
                semantic=(semantic.astype(int) - 1).reshape(h, w),
                H=h, W=w, direction_norms=nrm)
