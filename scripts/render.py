"""Render every frame of a fitted scene: RGB | depth | semantic | feature-PCA mosaic (CLI of the reference's
scripts/render.py).  Frames go to an .mp4 through skvideo when it is installed, otherwise to <out>/NNNNN.png."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from autolabel_amd import model_utils
from autolabel_amd.dataset import SceneDataset

COLORS = (np.array([[31, 119, 180], [255, 127, 14], [44, 160, 44], [214, 39, 40], [148, 103, 189], [140, 86, 75], [227, 119, 194],
                    [127, 127, 127], [188, 189, 34], [23, 190, 207]], dtype=np.uint8))  # matplotlib tab10


def read_args():
    parser = model_utils.model_flag_parser()
    parser.add_argument('scene')
    parser.add_argument('--fps', type=int, default=5)
    parser.add_argument('--stride', type=int, default=1)
    parser.add_argument('--model-dir', type=str, required=True)
    parser.add_argument('--max-depth', type=float, default=7.5)
    parser.add_argument('--out', type=str, required=True)
    return parser.parse_args()


def depth_colormap(depth, maxdepth):
    x = np.clip(depth / maxdepth, 0, 1)
    return (np.stack([x, 1 - np.abs(2 * x - 1), 1 - x], -1) * 255).astype(np.uint8)


def render(model, batch, size=(960, 720), maxdepth=10.0):
    rays_o = torch.tensor(batch['rays_o']).cuda()
    rays_d = torch.tensor(batch['rays_d']).cuda()
    direction_norms = torch.tensor(batch['direction_norms']).cuda()
    outputs = model.render(rays_o, rays_d, direction_norms, staged=True, perturb=False, num_steps=512, upsample_steps=0)
    sq = (size[0] // 2, size[1] // 2)
    frame = np.zeros((size[1], size[0], 3), dtype=np.uint8)
    frame[:sq[1], :sq[0]] = (outputs['image'].cpu().numpy() * 255.0).astype(np.uint8)
    frame[:sq[1], sq[0]:] = depth_colormap(outputs['depth'].cpu().numpy(), maxdepth)
    p_semantic = outputs['semantic'].argmax(dim=-1).cpu().numpy()
    frame[sq[1]:, :sq[0]] = COLORS[p_semantic % COLORS.shape[0]]
    f = outputs['semantic_features'].float()
    f = f.reshape(-1, f.shape[-1])
    _, _, V = torch.pca_lowrank(f - f.mean(0), q=3)
    pc = ((f - f.mean(0)) @ V[:, :3]).reshape(sq[1], sq[0], 3)
    pc = (pc - pc.amin((0, 1))) / (pc.amax((0, 1)) - pc.amin((0, 1)) + 1e-8)
    frame[sq[1]:, sq[0]:] = (pc.cpu().numpy() * 255).astype(np.uint8)
    return frame


def main():
    flags = read_args()
    params = model_utils.read_params(flags.model_dir)
    dataset = SceneDataset('test', flags.scene, size=(480, 360), batch_size=16384, features=params.features, load_semantic=False, lazy=True)
    n_classes = dataset.n_classes if dataset.n_classes is not None else 2
    kw = dict(cuda_ray=True, march_samples=getattr(params, 'march_samples', 96)) if getattr(params, 'cuda_ray', False) else {}
    model = model_utils.create_model(dataset.min_bounds, dataset.max_bounds, n_classes, params, **kw).cuda().eval()
    model_utils.load_checkpoint(model, os.path.join(flags.model_dir, 'checkpoints'))
    try:
        from skvideo.io.ffmpeg import FFmpegWriter
        writer = FFmpegWriter(flags.out, inputdict={'-framerate': f'{flags.fps}'},
                              outputdict={'-c:v': 'libx264', '-r': f'{flags.fps}', '-pix_fmt': 'yuv420p'})
    except ImportError:
        writer = None
        os.makedirs(flags.out, exist_ok=True)
    with torch.inference_mode():
        for i, frame_index in enumerate(dataset.indices[::flags.stride]):
            frame = render(model, dataset._get_test(frame_index), maxdepth=flags.max_depth)
            if writer is not None:
                writer.writeFrame(frame)
            else:
                from PIL import Image
                Image.fromarray(frame).save(os.path.join(flags.out, f'{i:05}.png'))
    if writer is not None:
        writer.close()


if __name__ == '__main__':
    main()
