"""Ray batches for NeRF training / rendering.

Two layers:

* the host-side mirror of the reference interface (``IndexSampler``, ``BaseDataset``, ``SceneDataset``,
  ``DynamicDataset``, ``LenDataset``, ``LazyImageLoader``, ``_compute_direction``) -- same names, batch dict keys,
  dtypes and RNG draw order as autolabel/dataset.py so scripts/train.py / render.py / export.py are drop-in;
* ``DeviceFrames`` -- the MI355X path: frames live in HBM and batches are assembled by the HIP ray-generation
  kernels (csrc/raygen.hip), removing the numpy/numba worker process + pickling + H2D copy of the reference
  (autolabel/dataset.py:182-242 -> autolabel/trainer.py:55-60).
"""
import ctypes as C
import os
import random
import threading
import time
from collections import deque

import numpy as np
import torch

_FLIP_YZ = np.diag([1.0, -1.0, -1.0, 1.0])  # OpenCV camera -> OpenGL camera


def nerf_matrix_to_ngp(pose, scale=1.0):
    """torch-ngp provider.nerf_matrix_to_ngp (external to the reference tree): rows (y,z,x), columns 1,2 negated."""
    p = np.asarray(pose)
    out = np.eye(4, dtype=np.float32)
    out[:3, 0] = p[[1, 2, 0], 0]
    out[:3, 1] = -p[[1, 2, 0], 1]
    out[:3, 2] = -p[[1, 2, 0], 2]
    out[:3, 3] = p[[1, 2, 0], 3] * scale
    return out


def convert_pose(T_CW):
    """world->camera (OpenCV) 4x4 to the camera->world pose in ngp axes (autolabel/dataset.py:268-274)."""
    return nerf_matrix_to_ngp(np.linalg.inv(T_CW) @ _FLIP_YZ, scale=1.0)


def _compute_direction(R_WC, ray_indices, w, fx, fy, cx, cy, randomize):
    """Host restatement of autolabel/dataset.py:17-37 (float64 pinhole division, float32 norm/rotation).
    Returns (directions [n,3] f32, norm [n,1] f32).  The device version is csrc/raygen.hip:pixel_direction."""
    ray_indices = np.asarray(ray_indices)
    n = ray_indices.size
    px = (ray_indices % w).astype(np.float32)
    py = ((ray_indices - px) / w).astype(np.float32)
    if randomize:
        px = px + np.random.random(n).astype(np.float32)
        py = py + np.random.random(n).astype(np.float32)
    else:
        px, py = px + np.float32(0.5), py + np.float32(0.5)
    d = np.ones((n, 3), dtype=np.float32)
    d[:, 0] = (px.astype(np.float64) - float(cx)) / float(fx)
    d[:, 1] = (py.astype(np.float64) - float(cy)) / float(fy)
    norm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])[:, None]
    d = d / norm
    R = np.asarray(R_WC, dtype=np.float32)
    out = np.stack([(R[r, 0] * d[:, 0] + R[r, 1] * d[:, 1]) + R[r, 2] * d[:, 2] for r in range(3)], 1)
    return out, norm


class LenDataset(torch.utils.data.IterableDataset):
    """Fixed-length view of an endless iterable dataset (one 'epoch' = `length` batches)."""

    def __init__(self, dataset, length):
        self.dataset, self.length = dataset, length

    def __iter__(self):
        it = iter(self.dataset)
        for _ in range(self.length):
            yield next(it)

    def __len__(self):
        return self.length


def _resize_nearest(img, size):
    """cv2.resize(..., INTER_NEAREST) replacement: src index = floor(dst * scale)."""
    w, h = size
    H, W = img.shape[:2]
    ys = np.minimum((np.arange(h) * (H / h)).astype(int), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(int), W - 1)
    return img[ys][:, xs]


def _resize_linear(img, size):
    """``cv2.resize(img, size)`` with the DEFAULT interpolation (INTER_LINEAR) on a single-channel uint16 / float32 image.

    This is what the reference's eager loader applies to depth frames: ``cv2.resize(depth_image, self.camera.size, cv2.INTER_NEAREST)``
    (autolabel/dataset.py:391-393) passes the flag in the ``dst`` position, so the interpolation argument keeps its default.  OpenCV is
    not in this image (parity unpinned); restated from its published algorithm (modules/imgproc/src/resize.cpp):
      * both scales exactly 2 (the default ``factor=2``): INTER_LINEAR is redirected to the fast INTER_AREA path, for 16-bit images
        ``(a + b + c + d + 2) >> 2`` over the 2 x 2 block (round half UP);
      * otherwise pixel-centre alignment ``fx = (dx + 0.5) * scale - 0.5`` in fp32, ``sx = floor(fx)``, clamped at both borders with
        weight 0, fp32 weights ``(1 - fx, fx)``, horizontal pass into fp32 rows, then the vertical blend, ``saturate_cast<ushort>`` =
        round half to EVEN and clamp to [0, 65535].
    """
    w, h = size
    H, W = img.shape[:2]
    if img.ndim != 2 or img.dtype not in (np.uint16, np.float32):
        raise NotImplementedError(f'_resize_linear: single-channel uint16 / float32 frames only (depth PNGs are 16-bit), got {img.dtype} {img.shape}')
    if (w, h) == (W, H):
        return img.copy()
    if img.dtype == np.uint16 and W == 2 * w and H == 2 * h:
        a = img.astype(np.uint32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint16)

    def taps(n_dst, n_src):
        scale = 1.0 / (n_dst / n_src)                        # (double, as cv::resize derives it from inv_scale)
        f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        s0[lo], f[lo] = 0, 0.0
        hi = s0 >= n_src - 1
        s0[hi], f[hi] = n_src - 1, 0.0
        s1 = np.minimum(s0 + 1, n_src - 1)
        return s0, s1, (np.float32(1.0) - f).astype(np.float32), f

    x0, x1, ax0, ax1 = taps(w, W)
    y0, y1, by0, by1 = taps(h, H)
    src = img.astype(np.float32)
    rows = src[:, x0] * ax0[None, :] + src[:, x1] * ax1[None, :]          # horizontal pass (fp32)
    out = rows[y0] * by0[:, None] + rows[y1] * by1[:, None]                # vertical blend (fp32)
    if img.dtype == np.float32:
        return out.astype(np.float32)
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)                # saturate_cast<ushort>: lrint (half to even) + clamp


class LazyImageLoader:
    """Decode + resize frames on first access, cache them (autolabel/dataset.py:55-77).  As in the reference EVERY frame -- depth
    included -- is converted to float32 and divided by 255 before the nearest-neighbour resize (its depth values in lazy mode are
    therefore raw / 255, bug for bug: autolabel/dataset.py:67)."""

    def __init__(self, images, size, interpolation=None):
        self.images, self.size, self._cache = images, size, {}

    def __getitem__(self, i):
        if i not in self._cache:
            from PIL import Image
            arr = np.array(Image.open(self.images[i]), dtype=np.float32) / 255.
            self._cache[i] = _resize_nearest(arr, self.size)
        return self._cache[i]

    def __len__(self):
        return len(self.images)

    @property
    def shape(self):
        return [len(self)]


class IndexSampler:
    """Per-class, per-image pixel index; images are drawn proportionally to their class pixel count
    (autolabel/dataset.py:80-151; pinned by the reference's test/test_sampling.py)."""

    def __init__(self):
        self.classes = np.array([])
        self.index = {}          # class id -> {image index -> pixel indices}
        self.image_weights = {}  # class id -> probability per image
        self.has_semantics = False
        self.image_range = np.array([])
        self.version = 0         # counts update() calls: every label-edit route of the reference ends in one (semantic_map_updated,
                                 # update_sampler), so a device-resident copy of the labels knows when it is stale (trainer.resident_loader)

    def update(self, semantic_maps):
        """0 is the null class, 1 background, 2.. object classes."""
        assert len(semantic_maps.shape) == 2
        self.version += 1
        labels = np.unique(semantic_maps)
        self.classes = labels[labels != 0]
        n = len(semantic_maps)
        self.index, counts = {}, {}
        for c in self.classes:
            per_image = (semantic_maps == c)
            n_pix = per_image.reshape(n, -1).sum(1)
            for i in np.flatnonzero(n_pix):
                self.has_semantics = True
                self.index.setdefault(c, {})[int(i)] = np.flatnonzero(per_image[i].ravel())
            if n_pix.sum() > 0:
                counts[c] = n_pix / n_pix.sum()
        self.image_weights = counts
        self.image_range = np.arange(n, dtype=int)

    def sample_class(self):
        return np.random.choice(self.classes)

    def sample(self, class_id, count=1):
        image_index = np.random.choice(self.image_range, p=self.image_weights[class_id])
        return image_index, np.random.choice(self.index[class_id][image_index], count)

    def semantic_indices(self):
        return sorted({i for per_class in self.index.values() for i in per_class})


class BaseDataset(torch.utils.data.IterableDataset):
    semantic_image_sample_ratio = 0.5

    def __init__(self, batch_size, camera):
        self.split = 'train'
        self.camera, self.batch_size = camera, batch_size
        self.pixel_indices = None
        self.features = None
        self.w, self.h = self.camera.size
        self.resolution = int(self.w * self.h)
        K = self.camera.camera_matrix
        self.intrinsics = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        self.sample_chunk_size = 512
        self.index_sampler = IndexSampler()

    def __iter__(self):
        if self.split == 'train':
            while True:
                yield self._next_train()
        else:
            for i in range(self.rotations.shape[0]):
                yield self._get_test(i)

    def _compute_direction(self, image_index, ray_indices, randomize=False):
        return _compute_direction(self.rotations[image_index], ray_indices, self.w, self.camera.fx, self.camera.fy,
                                  self.camera.cx, self.camera.cy, randomize)

    def _next_train(self):
        """One training batch = batch_size//512 chunks, each from ONE frame (dataset.py:182-242)."""
        cs = self.sample_chunk_size
        n_chunks = self.batch_size // cs
        B = n_chunks * cs
        out = {'rays_o': np.zeros((B, 3), np.float32), 'rays_d': np.zeros((B, 3), np.float32),
               'pixels': np.zeros((B, 3), np.float32), 'direction_norms': np.zeros((B, 1), np.float32),
               'depth': np.zeros(B, np.float32), 'semantic': np.zeros(B, dtype=int)}
        if self.features is not None:
            out['features'] = np.zeros((B, self.feature_dim), np.float32)
        for c in range(n_chunks):
            if self.index_sampler.has_semantics and random.random() < self.semantic_image_sample_ratio:
                frame, pix = self.index_sampler.sample(self.index_sampler.sample_class(), cs)
            else:
                frame = np.random.randint(0, self.n_examples)
                pix = np.random.choice(self.pixel_indices, size=(cs,))
            s = slice(c * cs, (c + 1) * cs)
            out['pixels'][s] = self.images[frame][pix]
            out['depth'][s] = self.depths[frame][pix] / 1000.0
            out['semantic'][s] = self.semantics[frame][pix].astype(int) - 1
            out['rays_o'][s] = self.origins[frame][None]
            out['rays_d'][s], out['direction_norms'][s] = self._compute_direction(frame, pix, randomize=True)
            if self.features is not None:
                x = pix % int(self.w)
                y = (pix - x) / int(self.w)
                fxy = self._scale_to_feature_xy(np.stack([x, y], -1))
                out['features'][s] = self.features[frame][fxy[:, 1] * self.feature_width + fxy[:, 0], :]
        return out

    def _get_test(self, image_index):
        """All rays of one frame at pixel centres (dataset.py:244-266; direction_norms stays [H*W,1])."""
        d, norms = self._compute_direction(image_index, np.arange(self.resolution))
        out = {'pixels': self.images[image_index].reshape(self.h, self.w, 3),
               'rays_o': np.broadcast_to(self.origins[image_index], (self.h, self.w, 3)).astype(np.float32),
               'rays_d': d.reshape(self.h, self.w, 3).astype(np.float32),
               'depth': (self.depths[image_index] / 1000.0).reshape(self.h, self.w),
               'semantic': (self.semantics[image_index].astype(int) - 1).reshape(self.h, self.w),
               'H': self.h, 'W': self.w, 'direction_norms': norms}
        if self.features is not None:
            out['features'] = self.features[image_index]
        return out

    def _convert_pose(self, T_CW):
        return convert_pose(T_CW)

    def _compute_rays(self):
        if self.split == 'train':
            self.images = self.images.reshape(self.n_examples, self.resolution, 3)
            self.depths = self.depths.reshape(self.n_examples, self.resolution)
            self.semantics = self.semantics.reshape(self.n_examples, self.resolution)

    def _compute_image_mask(self, images):
        """Pixels that are (almost) black in every sampled frame come from undistortion: never sample them."""
        if isinstance(images, LazyImageLoader):
            images = np.stack([images[i] for i in np.random.randint(0, len(images), size=5)])
        else:
            images = images[::10]
        lit = np.any(images > (10. / 255.), axis=-1)
        self.pixel_indices = np.flatnonzero(np.any(lit.reshape(lit.shape[0], -1), axis=0))

    def _set_feature_maps(self, features):
        N, Hf, Wf, Cf = features.shape
        self.features = features.reshape(N, Hf * Wf, Cf)
        self.feature_width, self.feature_height, self.feature_dim = Wf, Hf, Cf
        scale = np.array([Wf / self.camera.size[0], Hf / self.camera.size[1]])
        self._scale_to_feature_xy = lambda xy: (xy * scale).astype(int)

    def device_frames(self, device='cuda'):
        """Upload the frames once; batches are then generated on the GPU."""
        return DeviceFrames.from_dataset(self, device)


class SceneDataset(BaseDataset):
    """Scene directory reader (layout: reference README.md:107-135; autolabel/dataset.py:314-449)."""

    def __init__(self, split, scene, factor=4.0, size=None, batch_size=4096, lazy=False, features=None, load_semantic=True):
        from .utils import Scene
        self.lazy, self.load_semantic = lazy, load_semantic
        self.scene = Scene(scene)
        self.image_names = self.scene.image_names()
        full = self.scene.camera.size
        small = size if size is not None else (int(full[0] / factor), int(full[1] / factor))
        n = min(len(self.scene.rgb_paths()), len(self.scene.depth_paths()))
        self.indices = np.arange(0, n)
        super().__init__(batch_size, self.scene.camera.scale(small))
        self.split = split
        self._load_images()
        self._compute_rays()
        if features is not None:
            self._load_features(features)
        self.error_map = None
        self.n_classes = self.scene.n_classes

    def _load_images(self):
        from PIL import Image
        rgb_paths, depth_paths = self.scene.rgb_paths(), self.scene.depth_paths()
        images, depths, semantics, cameras = [], [], [], []
        for i in self.indices:
            if self.lazy:
                images.append(rgb_paths[i])
                depths.append(depth_paths[i])
            else:
                rgb = np.array(Image.open(rgb_paths[i]), dtype=np.float32)[..., :3]
                images.append(_resize_nearest(rgb, self.camera.size) / 255.)
                # (bilinear, not nearest: the reference passes cv2.INTER_NEAREST in the `dst` slot of cv2.resize -- autolabel/dataset.py:391-393)
                depths.append(_resize_linear(np.array(Image.open(depth_paths[i])), self.camera.size))
            sem_path = os.path.join(self.scene.path, 'semantic', os.path.basename(depth_paths[i]))
            if self.load_semantic and os.path.exists(sem_path):
                semantics.append(np.asarray(Image.open(sem_path).resize(self.camera.size, Image.NEAREST)))
            else:
                semantics.append(np.zeros(self.camera.size[::-1], dtype=np.uint8))
            cameras.append(self._convert_pose(self.scene.poses[i]).astype(np.float32))
        if self.lazy:
            self.images = LazyImageLoader(images, self.camera.size)
            self.depths = LazyImageLoader(depths, self.camera.size)
        else:
            self.images, self.depths = np.stack(images, 0), np.stack(depths, 0)
        self.semantics = np.stack(semantics)
        self.index_sampler.update(self.semantics.reshape(-1, self.resolution))
        self._compute_image_mask(self.images)
        self.poses = np.stack(cameras, 0)
        self.rotations = np.ascontiguousarray(self.poses[:, :3, :3])
        self.origins = self.poses[:, :3, 3]
        self.n_examples = self.images.shape[0]
        self.min_bounds, self.max_bounds = self.scene.bbox()

    def semantic_map_updated(self, image_index):
        from PIL import Image
        path = os.path.join(self.scene.path, 'semantic', f'{self.image_names[image_index]}.png')
        if not os.path.exists(path):
            print(f'Could not find image {path}')
            return
        image = np.asarray(Image.open(path).resize(self.camera.size, Image.NEAREST))
        self.semantics[image_index, :] = image.reshape(self.resolution)
        self.index_sampler.update(self.semantics)

    def update_sampler(self):
        self.index_sampler.update(self.semantics)

    def _load_features(self, name):
        """features.hdf 'features/<name>' [N,Hf,Wf,C] f16 (autolabel/dataset.py:438-449; written by
        scripts/compute_feature_maps.py:82-118).  Read by `utils.hdf5` (own HDF5 reader, pinned against h5py-written files);
        a sibling ``features_<name>.npy`` with the same array takes precedence when present."""
        npy = os.path.join(self.scene.path, f'features_{name}.npy')
        if os.path.exists(npy):
            self._set_feature_maps(np.load(npy))
            return
        from .utils import hdf5
        with hdf5.File(os.path.join(self.scene.path, 'features.hdf'), 'r') as hdf:
            self._set_feature_maps(hdf[f'features/{name}'][:])


class ArrayDataset(BaseDataset):
    """In-memory frames (synthetic scenes, tests) behind the same batch interface."""

    def __init__(self, scene, batch_size=4096, split='train'):
        from .utils import Camera
        fx, fy, cx, cy = scene['intrinsics']
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        super().__init__(batch_size, Camera(K, (scene['w'], scene['h'])))
        self.split = split
        to_np = lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        self.images = to_np(scene['images']).astype(np.float32)
        self.depths = to_np(scene['depths']).astype(np.uint16)
        self.semantics = to_np(scene['semantics']).astype(np.uint8)
        self.n_examples = self.images.shape[0]
        self.indices = np.arange(self.n_examples)
        self.poses = np.stack([convert_pose(T) for T in scene['T_CW']]).astype(np.float32)
        self.rotations = np.ascontiguousarray(self.poses[:, :3, :3])
        self.origins = self.poses[:, :3, 3]
        self.index_sampler.update(self.semantics)
        self._compute_image_mask(self.images.reshape(self.n_examples, self.h, self.w, 3))
        self.min_bounds, self.max_bounds = scene['min_bounds'], scene['max_bounds']
        self.n_classes = scene.get('n_classes')
        if scene.get('features') is not None:
            Hf, Wf = scene['feat_hw']
            f = to_np(scene['features'])
            self._set_feature_maps(f.reshape(self.n_examples, Hf, Wf, f.shape[-1]))


class DynamicDataset(BaseDataset):
    """Frames arrive online (ROS node); a background thread keeps a queue of batches (dataset.py:457-543)."""

    def __init__(self, batch_size, camera, capacity=None):
        super().__init__(batch_size, camera)
        self.capacity = capacity
        self.poses, self.rotations, self.origins = [], [], []
        self.images, self.depths, self.features, self.semantics = [], [], [], []
        self.n_examples = 0
        self.prefetch_buffer, self.prefetch_buffer_size = deque(), 25
        self.stopped = False
        self._lock = threading.Lock()
        self._prefetch_thread = threading.Thread(target=self._prefetch, daemon=True)
        self._prefetch_thread.start()

    def stop(self):
        self.stopped = True
        self._prefetch_thread.join()

    def _prefetch(self):
        while not self.stopped:
            if len(self.features) == 0 or len(self.prefetch_buffer) >= self.prefetch_buffer_size:
                time.sleep(0.1)
                continue
            with self._lock:
                batch = self._next_train()
            self.prefetch_buffer.append(batch)

    def __iter__(self):
        while True:
            if len(self.prefetch_buffer) == 0:
                time.sleep(0.1)
            else:
                yield self.prefetch_buffer.popleft()

    def __len__(self):
        return self.n_examples

    def add_frame(self, T_CW, rgb, depth, features):
        assert depth.dtype == np.uint16 and rgb.dtype == np.uint8 and len(features.shape) == 3
        with self._lock:
            if len(self.features) == 0:
                self.feature_height, self.feature_width, self.feature_dim = features.shape
                scale = np.array([self.feature_width / self.camera.size[0], self.feature_height / self.camera.size[1]])
                self._scale_to_feature_xy = lambda xy: (xy * scale).astype(int)
            assert features.shape[0] == self.feature_height
            if self.pixel_indices is None:
                self.resolution = rgb.shape[0] * rgb.shape[1]
                self.pixel_indices = np.arange(self.resolution)
            T_WC = self._convert_pose(T_CW)
            columns = (self.poses, self.rotations, self.origins, self.images, self.depths, self.features, self.semantics)
            values = (T_WC, np.ascontiguousarray(T_WC[:3, :3]), T_WC[:3, 3], rgb.reshape(-1, 3) / 255., depth.reshape(-1),
                      features.reshape(self.feature_height * self.feature_width, features.shape[2]),
                      np.zeros(self.resolution, dtype=np.uint16))
            for col, v in zip(columns, values):
                col.append(v)
            if self.capacity is not None and len(self.poses) > self.capacity:
                drop = np.random.randint(0, len(self.poses))
                for col in columns:
                    del col[drop]
            self.n_examples = len(self.images)


class DeviceFrames:
    """Frames resident in HBM + HIP batch assembly (the accelerated R1-R3 rows of SURVEY.md section 8a)."""

    def __init__(self, images, depths, semantics, poses, pixel_indices, w, h, intrinsics, features=None, feat_hw=None,
                 device='cuda'):
        from . import hip as H
        H.require_gpu()
        self.H = H
        dv = torch.device(device)
        t = lambda a, dt: torch.as_tensor(a).to(device=dv, dtype=dt).contiguous()
        self.images = t(images, torch.float32)
        self.depths = torch.as_tensor(np.asarray(depths.cpu() if torch.is_tensor(depths) else depths).astype(np.uint16)
                                      .view(np.int16)).to(dv).contiguous()
        self.semantics = t(semantics, torch.uint8)
        poses = np.asarray(poses, dtype=np.float32)
        self.rotations = t(np.ascontiguousarray(poses[:, :3, :3]), torch.float32)
        self.origins = t(np.ascontiguousarray(poses[:, :3, 3]), torch.float32)
        self.pixel_indices = t(np.asarray(pixel_indices), torch.int32)
        self.n_frames, self.w, self.h = self.images.shape[0], int(w), int(h)
        self.features = None
        self.feature_dim = 0
        fr = self.desc = H.AlnFrames()
        fr.images, fr.depths, fr.semantics = self.images.data_ptr(), self.depths.data_ptr(), self.semantics.data_ptr()
        fr.rotations, fr.origins, fr.pixel_indices = self.rotations.data_ptr(), self.origins.data_ptr(), self.pixel_indices.data_ptr()
        fr.n_frames, fr.w, fr.h, fr.n_pix = self.n_frames, self.w, self.h, self.pixel_indices.numel()
        fr.fx, fr.fy, fr.cx, fr.cy = [float(v) for v in intrinsics]
        if features is not None:
            self.features = t(features, torch.float16)
            fr.features = self.features.data_ptr()
            fr.feat_h, fr.feat_w, fr.feat_c = int(feat_hw[0]), int(feat_hw[1]), int(self.features.shape[-1])
            self.feature_dim = int(self.features.shape[-1])
        self.device = dv
        self.set_class_index(np.asarray(semantics.cpu() if torch.is_tensor(semantics) else semantics))

    def set_class_index(self, semantics, ratio=BaseDataset.semantic_image_sample_ratio):
        """Device copy of IndexSampler (autolabel/dataset.py:80-151): per class, per frame pixel lists in CSR form.
        Call again after the labels change (SceneDataset.update_sampler)."""
        sem = np.asarray(semantics).reshape(self.n_frames, -1)
        classes = np.unique(sem)
        classes = classes[classes != 0]
        fr = self.desc
        # the descriptor is passed BY VALUE into aln_raygen_train: a captured step keeps the old pointers / class count, so whoever
        # replays one compares `version` (engine.TrainEngine.graphed's guard)
        self.version = getattr(self, 'version', 0) + 1
        fr.n_classes, fr.sem_ratio = 0, 0.0
        if len(classes) == 0:
            return
        offs, pix, base = [], [], 0
        for c in classes:
            row = [base]
            for f in range(self.n_frames):
                idx = np.flatnonzero(sem[f] == c)
                pix.append(idx)
                base += len(idx)
                row.append(base)
            offs.append(row)
        self.cls_offsets = torch.as_tensor(np.asarray(offs, dtype=np.int32)).to(self.device).contiguous()
        self.cls_pixels = torch.as_tensor(np.concatenate(pix).astype(np.int32)).to(self.device).contiguous()
        self.classes = classes
        fr.cls_offsets, fr.cls_pixels = self.cls_offsets.data_ptr(), self.cls_pixels.data_ptr()
        fr.n_classes, fr.sem_ratio = len(classes), float(ratio)

    @classmethod
    def from_dataset(cls, ds, device='cuda'):
        feats = ds.features if ds.features is not None else None
        fhw = (ds.feature_height, ds.feature_width) if feats is not None else None
        return cls(ds.images, ds.depths, ds.semantics, ds.poses, ds.pixel_indices, ds.w, ds.h,
                   (ds.camera.fx, ds.camera.fy, ds.camera.cx, ds.camera.cy), feats, fhw, device)

    @classmethod
    def from_scene(cls, scene, device='cuda'):
        poses = np.stack([convert_pose(T) for T in scene['T_CW']])
        n_pix = scene['w'] * scene['h']
        return cls(scene['images'], scene['depths'], scene['semantics'], poses, np.arange(n_pix), scene['w'], scene['h'],
                   scene['intrinsics'], scene.get('features'), scene.get('feat_hw'), device)

    def world_to_camera(self):
        """[F,4,4] T_CW in the renderer's frame (inverse of the converted poses): input of mark_untrained_grid."""
        F = self.n_frames
        T = torch.eye(4, dtype=torch.float64).repeat(F, 1, 1)
        T[:, :3, :3] = self.rotations.double().cpu()
        T[:, :3, 3] = self.origins.double().cpu()
        return torch.linalg.inv(T).float().numpy()

    def alloc_batch(self, B):
        dv = self.device
        b = {'rays_o': torch.empty(B, 3, device=dv), 'rays_d': torch.empty(B, 3, device=dv),
             'direction_norms': torch.empty(B, 1, device=dv), 'pixels': torch.empty(B, 3, device=dv),
             'depth': torch.empty(B, device=dv), 'semantic': torch.empty(B, dtype=torch.int32, device=dv)}
        if self.features is not None:
            b['features'] = torch.empty(B, self.feature_dim, device=dv)
        return b

    def _batch_desc(self, b):
        d = self.H.AlnBatch()
        d.rays_o, d.rays_d, d.norms = b['rays_o'].data_ptr(), b['rays_d'].data_ptr(), b['direction_norms'].data_ptr()
        d.pixels, d.depth, d.semantic = b['pixels'].data_ptr(), b['depth'].data_ptr(), b['semantic'].data_ptr()
        d.features = b['features'].data_ptr() if 'features' in b else None
        return d

    def next_train(self, out, seed, step, frame_range=None, chunk=512, chunk_frames=None, ray_idx=None, step_dev=None):
        """Fill `out` (from alloc_batch) with one training batch; frame_range=(lo,hi) shards frames across ranks.
        step_dev: optional device int32[1] added to `step` inside the kernel (hipGraph replays: engine.GraphedStep)."""
        H = self.H
        B = out['rays_o'].shape[0]
        lo, hi = frame_range if frame_range is not None else (0, self.n_frames)
        H.call('aln_raygen_train', C.byref(self.desc), C.byref(self._batch_desc(out)), B, chunk, lo, hi, seed, step,
               H.ptr(chunk_frames), H.ptr(ray_idx), None, H.ptr(step_dev), H.stream())
        return out

    def get_test(self, frame, out=None):
        H = self.H
        out = out if out is not None else self.alloc_batch(self.w * self.h)
        H.call('aln_raygen_frame', C.byref(self.desc), C.byref(self._batch_desc(out)), int(frame), H.stream())
        return out


class DeviceLoader:
    """Endless iterable of device batches: frames resident in HBM, batches assembled by the HIP ray-generation kernel
    (replaces DataLoader + worker process + H2D copy of scripts/train.py:64-68).  ``frame_range`` = this rank's shard of the
    frames under data parallelism (parallel.frame_shard); ``seed`` should differ per rank (parallel.rank_seed)."""

    def __init__(self, frames, batch_size, length, seed=0, frame_range=None):
        self.frames, self.batch, self.length = frames, frames.alloc_batch(batch_size), length
        self.seed, self.step, self.frame_range = seed, 0, frame_range

    def __iter__(self):
        for _ in range(self.length):
            self.frames.next_train(self.batch, self.seed, self.step, frame_range=self.frame_range)
            self.step += 1
            yield self.batch

    def __len__(self):
        return self.length
