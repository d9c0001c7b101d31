"""Held-out quality of a trained field: PSNR, depth L1 and mIoU on frames the optimiser never saw.

The headline metric is "rays/s at fixed PSNR / mIoU" (BASELINE.json), so throughput numbers travel with these.
mIoU follows the reference's definition (autolabel/evaluation.py:21-29 `compute_iou`: intersection over union of the
argmax of the rendered logits, classes 1.. = every labelled class, scripts/evaluate.py:100-105 takes the mean), accumulated
over the held-out frames.  `render` is any callable ``(rays_o, rays_d, direction_norms) -> dict`` with the keys of
``ALNetwork.render`` -- the HIP model in the product, the CPU oracle in the parity tests -- so both sides of a
comparison go through the same metric code.
"""
import math

import numpy as np
import torch

DEPTH_EPSILON = 0.01   # autolabel/trainer.py:13


def split_heldout(n_frames, every=20, first=5):
    held = list(range(first, n_frames, every))
    return [i for i in range(n_frames) if i not in held], held


def frame_metrics(out, pixels, depth, semantic, n_classes, inter, union):
    """Accumulates one frame; returns (psnr, depth_l1).  semantic: int labels, -1 = unlabeled."""
    img = out['image'].reshape(-1, 3).float().cpu()
    px = pixels.reshape(-1, 3).float().cpu()
    mse = ((img - px) ** 2).mean().item()
    d, gd = out['depth'].reshape(-1).float().cpu(), depth.reshape(-1).float().cpu()
    valid = gd > DEPTH_EPSILON
    l1 = (d[valid] - gd[valid]).abs().mean().item() if valid.any() else float('nan')
    pred = out['semantic'].reshape(-1, out['semantic'].shape[-1]).argmax(-1).cpu().numpy()
    gt = semantic.reshape(-1).cpu().numpy()
    lab = gt >= 0
    for c in range(n_classes):
        inter[c] += ((pred == c) & (gt == c) & lab).sum()
        union[c] += (((pred == c) | (gt == c)) & lab).sum()
    return -10.0 * math.log10(max(mse, 1e-12)), l1


def heldout_metrics(render, test_frames, n_classes, frames=None):
    """test_frames: dataset.DeviceFrames (or anything with .n_frames and .get_test(i) -> batch dict of tensors)."""
    inter, union = np.zeros(n_classes), np.zeros(n_classes)
    psnr, l1 = [], []
    with torch.no_grad():
        for f in (frames if frames is not None else range(test_frames.n_frames)):
            t = test_frames.get_test(f)
            out = render(t['rays_o'], t['rays_d'], t['direction_norms'])
            p, d = frame_metrics(out, t['pixels'], t['depth'], t['semantic'], n_classes, inter, union)
            psnr.append(p); l1.append(d)
    seen = union > 0
    return {'psnr_db': float(np.mean(psnr)), 'depth_l1_m': float(np.nanmean(l1)),
            'miou': float((inter[seen] / union[seen]).mean()) if seen.any() else float('nan'),
            'frames': len(psnr), 'classes_scored': int(seen.sum())}


def pipe_renderer(pipe, num_steps=256, upsample_steps=0, chunk=16384):
    """`render` callable on a bare pipeline.HipPipeline (what bench.py trains), same settings as scripts/quality.py."""
    def render(rays_o, rays_d, norms):
        ro, rd, dn = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), norms.reshape(-1)
        parts = []
        for a in range(0, ro.shape[0], chunk):
            out, _ = pipe.forward(ro[a:a + chunk].contiguous(), rd[a:a + chunk].contiguous(), dn[a:a + chunk].contiguous(),
                                  num_steps, upsample_steps, False, train=False)
            parts.append({k: out[k].clone() for k in ('image', 'depth', 'semantic')})
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
    return render
