"""Quality check of the HIP training path on the synthetic room scene: PSNR on held-out frames and mIoU of the rendered
semantic argmax against the full ground-truth labels (only every 10th frame is labelled for training).

    python scripts/quality.py --iters 3000
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import autolabel_amd  # noqa: F401  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before HIP initialises)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=3000)
    ap.add_argument('--frames', type=int, default=100)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--cuda-ray', action='store_true', help='occupancy-grid marching (csrc/march.hip)')
    ap.add_argument('--march-samples', type=int, default=64)
    args = ap.parse_args()
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.models import ALNetwork
    dev = 'cuda'
    scene = synthetic.subsample(synthetic.make_room_scene(n_frames=args.frames, device=dev, feat_dim=64, seed=1), 2)
    held = list(range(5, args.frames, 20))
    train_ids = [i for i in range(args.frames) if i not in held]
    sub = lambda k, ids: scene[k][ids]
    tr = dict(scene, images=sub('images', train_ids), depths=sub('depths', train_ids), semantics=sub('semantics', train_ids),
              features=sub('features', train_ids), T_CW=scene['T_CW'][train_ids])
    frames = DeviceFrames.from_scene(tr, dev)
    test = DeviceFrames.from_scene(dict(scene, images=sub('images', held), depths=sub('depths', held), semantics=sub('semantics_full', held),
                                        features=None, T_CW=scene['T_CW'][held]), dev)
    lo, hi = scene['min_bounds'], scene['max_bounds']
    bound = float(((hi - lo) - (lo + hi) * 0.5).max())
    model = ALNetwork(encoding='hg+freq', num_layers=2, hidden_dim=128, geo_feat_dim=15, num_layers_color=2, hidden_dim_color=128,
                      hidden_dim_semantic=64, semantic_classes=scene['n_classes'], bound=bound, cuda_ray=args.cuda_ray, density_scale=1,
                      march_samples=args.march_samples).cuda()
    if args.cuda_ray:
        model._ensure_device().mark_untrained_grid(frames.world_to_camera(), (frames.desc.fx, frames.desc.fy, frames.desc.cx, frames.desc.cy),
                                                   size=(frames.w, frames.h))
    eng = TrainEngine(model._ensure_device(), feature_loss=True)
    batch = frames.alloc_batch(args.batch)
    gamma, steps = 0.5, math.log(1e-4 / 5e-3, 0.5)
    step_size = max(args.iters // steps // 1000, 1) * 1000  # StepLR of scripts/train.py:70-75, applied per 1000 iterations
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(args.iters):
        eng.lr = 5e-3 * gamma ** (i // step_size)
        frames.next_train(batch, seed=3, step=i)
        eng.step(batch, seed=4, step=i)
    torch.cuda.synchronize(); dt = time.time() - t0
    model.eval()
    psnrs, inter, union, depth_err = [], np.zeros(8), np.zeros(8), []
    with torch.inference_mode():
        for f in range(len(held)):
            t = test.get_test(f)
            if args.cuda_ray:
                model.march_samples = 128
            out = model.render(t['rays_o'], t['rays_d'], t['direction_norms'], staged=True, perturb=False, num_steps=256,
                               upsample_steps=0, max_ray_batch=16384)
            mse = ((out['image'] - t['pixels']) ** 2).mean().item()
            psnrs.append(-10 * math.log10(mse))
            valid = t['depth'] > 0.01
            depth_err.append((out['depth'][valid] - t['depth'][valid]).abs().mean().item())
            pred = out['semantic'].argmax(-1).cpu().numpy()
            gt = t['semantic'].cpu().numpy()
            for c in range(scene['n_classes']):
                inter[c] += ((pred == c) & (gt == c)).sum(); union[c] += ((pred == c) | (gt == c)).sum()
    iou = inter[union > 0] / union[union > 0]
    print('per-class IoU', np.round(inter / np.maximum(union, 1), 3).tolist(), 'pred hist', np.bincount(pred.ravel(), minlength=8).tolist(), 'gt hist', np.bincount(gt.ravel(), minlength=8).tolist(), file=sys.stderr)
    print(json.dumps({'iters': args.iters, 'cuda_ray': args.cuda_ray, 'samples_per_ray': args.march_samples if args.cuda_ray else 256, 'train_rays_per_s': args.batch * args.iters / dt, 'psnr_heldout': float(np.mean(psnrs)),
                      'depth_l1_m': float(np.mean(depth_err)), 'miou_heldout': float(iou.mean()), 'loss_terms': eng.terms.tolist(),
                      'loss_scale': eng.state_f[0].item(), 'adam_steps': int(eng.state_i[0].item())}))


if __name__ == '__main__':
    main()
