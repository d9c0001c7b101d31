import os, sys, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
def _worker(rank, world, port, mode):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo'); torch.cuda.set_device(0)
    from autolabel_amd import parallel, synthetic
    from autolabel_amd.dataset import DeviceFrames
    from autolabel_amd.engine import TrainEngine
    from autolabel_amd.pipeline import HipPipeline, ModelLayout, Params
    dev = torch.device('cuda', 0)
    scene = synthetic.make_cube_scene(n_frames=8, size=32, seed=0, device=dev)
    frames = DeviceFrames.from_scene(scene, dev)
    layout = ModelLayout('hg+freq', 15, 128, 128, 64, scene['n_classes'], bound=3.0)
    P = Params(layout, dev); P.init_(seed=0); parallel.broadcast_parameters(P.flat); P.refresh_shadows()
    eng = TrainEngine(HipPipeline(layout, P), process_group=dist.group.WORLD, num_steps=32, upsample_steps=32)
    if mode == 'mainstream':
        eng._comm = torch.cuda.current_stream()
    lo, hi = parallel.frame_shard(8, rank, world)
    batch = frames.alloc_batch(1024)
    for i in range(3):
        frames.next_train(batch, seed=parallel.rank_seed(5, rank), step=i, frame_range=(lo, hi))
        eng.forward_backward(batch, seed=parallel.rank_seed(7, rank), step=i)
        eng.all_reduce_grads()
        torch.cuda.synchronize()
        g = P.grad.detach().cpu().clone()
        both = [None] * world
        dist.all_gather_object(both, g)
        if rank == 0:
            d = (both[0] - both[1]).abs()
            ng = layout.n_grid
            print(mode, 'grad diff grid', d[:ng].max().item(), 'mlp', d[ng:layout.n_total].max().item(), 'tail', both[0][layout.n_total:].tolist(), both[1][layout.n_total:].tolist())
            F = 2
            for lo_, hi_ in eng.level_groups():
                a = int(layout.enc.grid.offset[lo_]) * F; b = int(layout.enc.grid.offset[hi_]) * F if hi_ < 16 else ng
                print('  levels', lo_, hi_, a, b, d[a:b].max().item(), 'nonzero frac', (both[0][a:b] != 0).float().mean().item())
        eng.optimizer_step()
        torch.cuda.synchronize()
        fl = P.flat.detach().cpu().clone(); bothp = [None] * world; dist.all_gather_object(bothp, fl)
        if rank == 0: print('  step', i, 'param diff', (bothp[0] - bothp[1]).abs().max().item(), 'state_i', eng.state_i[:8].tolist())
    dist.barrier(); dist.destroy_process_group()
if __name__ == '__main__':
    for mode in ['overlap', 'mainstream']:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_worker, args=(2, port, mode), nprocs=2, join=True)
