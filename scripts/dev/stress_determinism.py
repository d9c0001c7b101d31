"""Is one training step a pure function of (parameters, batch, seeds)?  VERDICT r4 item 1.

Runs the SAME forward + loss + backward of the default model again and again from the same state and compares every gradient block,
every output and the step's intermediates bit for bit with the first run, while (optionally)
  --poison    every workspace / slab / staging buffer is refilled with fresh random bytes before each run (a kernel that reads scratch
              it did not write shows up as a mismatch that follows the garbage), and
  --disturb   a second process runs training steps of its own on the same GPU (the condition of tests/test_gpu_parallel.py: blocks of
              two processes interleave on the CUs, so an intra-kernel race that lock-step scheduling hides gets its chance).
  --groups    the hash-grid scatter runs in the data-parallel level groups on odd iterations (must equal the single launch).
Prints one line per buffer that ever differed and exits 1 if any gradient / output did.
"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch


def build(batch_rays, S, seed=0, classes=None, size=32):
    from autolabel_amd import synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    scene = synthetic.make_cube_scene(n_frames=8, size=size, seed=0, device=dev)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, classes or scene['n_classes'], bound=3.0)
    P = Params(layout, dev); P.init_(seed=seed)
    with torch.no_grad():
        P.flat[:layout.n_grid].mul_(2e3)      # a table that shapes the density, so that the colour head sees a live / dead mix
    P.refresh_shadows()
    eng = TrainEngine(HipPipeline(layout, P), num_steps=S, upsample_steps=S, fuse_grid_adam=False)
    batch = frames.alloc_batch(batch_rays)
    frames.next_train(batch, seed=5, step=0)
    return eng, P, layout, batch, frames


def poison_all(eng, P, gen_seed):
    g = torch.Generator(device='cuda').manual_seed(gen_seed)
    def fill(t):
        if t is None or t.numel() == 0:
            return
        b = t.view(torch.uint8) if t.is_contiguous() else None
        if b is not None:
            b.random_(0, 256, generator=g)
    for ws in (eng.ws, eng.pipe.ws):
        for key, t in ws.bufs.values():
            fill(t)
    for k, fr in P.frags.items():
        fill(fr[3])
    for name, t in eng._g.items():
        fill(t)


def snapshot(eng, P, layout, out):
    snap = {}
    L = layout
    F, g = 2, L.enc.grid
    for l in range(int(g.n_levels)):
        a = int(g.offset[l]) * F
        b = int(g.offset[l + 1]) * F if l + 1 < int(g.n_levels) else L.n_grid
        snap[f'grad/level{l:02d}'] = P.grad[a:b].clone()
    for k in L.nets:
        snap[f'grad/{k}'] = P.net_view(k, P.grad).clone()
    for k, v in out.items():
        snap[f'out/{k}'] = v.clone()
    snap['terms'] = eng.terms.clone()
    snap['counts'] = eng.counts[:2].clone()
    snap['found_inf'] = eng.state_i[2:4].clone()
    for name in ('z', 'sigma', 'sigma_out', 'w_row', 'perm', 'n_live', 'sem_tile_sums', 'sem_dots', 'd_h0', 'd_semf_in', 'd_sigma_out', 'd_enc', 'enc'):
        t = eng.ws.bufs.get(name)
        if t is not None:
            snap[f'ws/{name}'] = t[1].clone()
    for name in ('g_image', 'g_depth', 'g_sem', 'g_feat'):
        if name in eng._g:
            snap[f'g/{name}'] = eng._g[name].clone()
    return snap


def bits(t):
    t = t.contiguous()
    return t.view(torch.uint8) if t.dtype != torch.bool else t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--S', type=int, default=32)
    ap.add_argument('--poison', action='store_true')
    ap.add_argument('--disturb', action='store_true')
    ap.add_argument('--groups', action='store_true')
    ap.add_argument('--role', default='main')
    ap.add_argument('--seconds', type=float, default=60.0)
    ap.add_argument('--classes', type=int, default=0)
    ap.add_argument('--trace', default='')
    a = ap.parse_args()
    import autolabel_amd  # noqa: F401  (sets the graph-capture workaround before HIP comes up)
    eng, P, layout, batch, frames = build(a.batch, a.S, classes=a.classes or None)
    if a.role == 'disturb':     # the competing process: plain training steps until told to stop
        t0 = time.time()
        i = 0
        if a.trace:      # every C-ABI call logged (flushed) and synchronised: the last line of the log names the faulting launch
            from autolabel_amd import hip as H
            log = open(a.trace, 'w')
            inner = H.call
            def traced(name, *args):
                log.write(f'{i} {name}\n'); log.flush()
                inner(name, *args)
                torch.cuda.synchronize()
            H.call = traced
        while time.time() - t0 < a.seconds:
            frames.next_train(batch, seed=9, step=i)
            eng.step(batch, seed=11, step=i)
            i += 1
            nl = eng.ws.bufs['n_live'][1]
            zero_live = nl.clone() if i == 1 else torch.minimum(zero_live, nl)
            if i % 20 == 0:
                torch.cuda.synchronize()
            if a.trace and i % 200 == 0:
                log.write(f'# step {i} n_live {int(eng.ws.bufs["n_live"][1].item())} finite {bool(torch.isfinite(P.flat).all())} '
                          f'state_i {eng.state_i[:5].tolist()} scale {float(eng.state_f[0])} terms {eng.terms.tolist()}\n')
        torch.cuda.synchronize()
        print('disturb role: finished', i, 'steps; smallest n_live seen', int(zero_live.item()), flush=True)
        return 0
    child = None
    if a.disturb:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--role', 'disturb', '--batch', '4096', '--S', '64', '--seconds', str(a.seconds)])
        time.sleep(20.0)     # let it come up (first import of torch on a fresh box)
    ref, bad, ever = None, {}, 0
    groups = eng.level_groups()
    for it in range(a.iters):
        P.grad.zero_(); eng.state_i[2:4] = 0
        if a.poison:
            poison_all(eng, P, 1000 + it)
        torch.cuda.synchronize()
        if a.groups and it % 2 == 1:
            # what TrainEngine.forward_backward does under data parallelism with overlap, minus the collectives
            eng.world, eng.overlap_comm = 2, True
            eng.direct_wire = False      # (keep the table's gradient in P.grad, where this tool compares it)
            if hasattr(eng, 'dp'):
                eng.dp = True
            saved = eng._bucket_ready
            eng._bucket_ready = lambda kind, x, y: None
            out = eng.forward_backward(batch, seed=7, step=0)
            eng._bucket_ready = saved
            eng.world, eng.overlap_comm = 1, False
            if hasattr(eng, 'dp'):
                eng.dp = False
        else:
            out = eng.forward_backward(batch, seed=7, step=0)
        torch.cuda.synchronize()
        snap = snapshot(eng, P, layout, out)
        if ref is None:
            ref = snap
            continue
        for k, v in snap.items():
            if k == 'found_inf' and a.groups:
                continue
            r = ref[k]
            if r.shape != v.shape or not torch.equal(bits(r), bits(v)):
                n = int((r != v).sum().item()) if r.shape == v.shape else -1
                bad.setdefault(k, []).append((it, n))
    if child is not None:
        child.wait()
    hard = [k for k in bad if k.startswith(('grad/', 'out/', 'terms', 'counts'))]
    for k in sorted(bad):
        its = bad[k]
        print(f'{"HARD" if k in hard else "soft"} {k}: differed in {len(its)}/{a.iters - 1} runs; first (iter, elements) {its[:4]}')
    print(f'stress_determinism poison={a.poison} disturb={a.disturb} groups={a.groups} batch={a.batch} S={a.S}: '
          f'{"MISMATCH in " + ", ".join(sorted(hard)) if hard else "all gradients and outputs bit-identical"} over {a.iters} runs; n_live={int(ref["ws/n_live"].item())}')
    return 1 if hard else 0


if __name__ == '__main__':
    sys.exit(main())
