"""Open-vocabulary queries on a trained feature field.

The reference answers "which text prompt does this pixel / 3-D point look like" in ``autolabel/evaluation.py``:

* 2-D (``OpenVocabEvaluator._predict_semantic``, evaluation.py:295-327): render the semantic features of a frame, normalise
  each pixel's feature, take the dot product with every text feature in a Python loop over image rows, argmax.
* 3-D (``OpenVocabEvaluator3D._predict_semantic``, evaluation.py:400-445): ``density`` + ``semantic`` point queries in
  batches of 50 000, averaged over 10 evaluations (the point itself plus 9 copies jittered with N(0, 0.02)), normalise,
  one dot product per prompt in a Python loop, argmax.

Here the point path is one launch sequence per jitter (hash-grid encode -> sigma head -> both semantic heads on the HIP
kernels, no geo_feat round trip through fp32) and the prompt comparison is one kernel (``aln_similarity_argmax``: fp32 dot
products against the prompt matrix and the argmax, no [n, C] similarity tensor).
"""
import ctypes as C

import torch

from . import hip as H
from .pipeline import f16


def similarity_argmax(features, text_features):
    """argmax_c <f / |f|, t_c> for every row of ``features`` [n, D] against ``text_features`` [C, D] -> int64 [n]
    (``aln_similarity_argmax``, csrc/heads.hip: fp32 dot products, first maximum).

    The division by the feature norm does not change the argmax, but rows with a zero feature are NaN in the reference
    and argmax to class 0 there (evaluation.py:304, 426); that is kept."""
    f = features.reshape(-1, features.shape[-1]).float().contiguous()
    t = text_features.float().to(f.device).contiguous()
    assert t.shape[1] == f.shape[1], 'text features and rendered features must have the same width'
    out = torch.empty(f.shape[0], dtype=torch.int64, device=f.device)
    H.call('aln_similarity_argmax', H.ptr(f), f.shape[0], f.shape[1], H.ptr(t), t.shape[0], H.ptr(out), H.stream())
    return out.reshape(features.shape[:-1])


@torch.no_grad()
def point_features(model, points):
    """Semantic features [n, D] (fp32) of 3-D points: ``model.semantic(model.density(x)['geo_feat'])[1]`` in one launch
    sequence (models.py:175-188, 248-256)."""
    pipe, L = model._ensure_device(), model._layout
    sigma_out, _ = model._sigma_rows(points)
    n = sigma_out.shape[0]
    fs, os_ = L.nets['semf'], L.nets['semo']
    if L.sem_wide:   # wide (LSeg) heads: model.semantic runs them on the wide.hip GEMMs
        return model.semantic(sigma_out[:, 1:1 + L.G])[1].float()
    feat = torch.empty(n, fs.out_pad, dtype=f16, device=sigma_out.device)
    logits = torch.empty(n, os_.out_pad, dtype=f16, device=sigma_out.device)
    H.call('aln_sem_heads_fwd', C.byref(pipe.P.descs['semf']), C.byref(pipe.P.descs['semo']), H.ptr(sigma_out), n, L.D, L.G,
           H.ptr(feat), H.ptr(logits), H.stream())
    return feat[:, :L.D].float()


@torch.no_grad()
def predict_semantic_points(model, points, text_features, label_id_map=None, n_evals=10, jitter_std=0.02, batch_size=50000,
                            generator=None):
    """evaluation.py:400-445: feature of the point + (1 / n_evals) x features of n_evals - 1 jittered copies, prompt argmax.

    (The reference adds the un-jittered feature with weight 1 and the jittered ones with weight 1 / n_evals; that
    weighting is kept.)  ``generator`` seeds the jitter; the reference uses the global torch RNG."""
    pts = points.reshape(-1, 3).float()
    out = torch.empty(pts.shape[0], dtype=torch.int64, device=pts.device)
    scale = 1.0 / n_evals
    for a in range(0, pts.shape[0], batch_size):
        batch = pts[a:a + batch_size]
        feats = point_features(model, batch)
        for _ in range(n_evals - 1):
            noise = torch.randn(batch.shape, device=batch.device, dtype=batch.dtype, generator=generator) * jitter_std
            feats += point_features(model, batch + noise) * scale
        out[a:a + batch_size] = similarity_argmax(feats, text_features)
    return label_id_map[out] if label_id_map is not None else out


@torch.no_grad()
def predict_semantic_image(model, batch, text_features, label_id_map=None, **render_kwargs):
    """evaluation.py:295-327: staged render of one frame (``batch`` = ``get_test`` dict), per-pixel prompt argmax [H, W]."""
    dev = next(model.parameters()).device
    as_t = lambda v: (v if torch.is_tensor(v) else torch.as_tensor(v)).to(dev)
    outputs = model.render(as_t(batch['rays_o']), as_t(batch['rays_d']), as_t(batch['direction_norms']), staged=True,
                           perturb=False, **render_kwargs)
    idx = similarity_argmax(outputs['semantic_features'], text_features)
    return label_id_map[idx] if label_id_map is not None else idx
