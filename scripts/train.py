"""Fit a scene (CLI of the reference's scripts/train.py).  `--synthetic room|cube` trains on a generated scene instead
of a directory.  The frames are kept in HBM and the batches assembled by the HIP ray-generation kernels whenever they fit (they
do for every scene of the reference's size: 200 frames at 320x240 are 230 MB); `--host-data` keeps the reference's host loader
(DataLoader + worker process + H2D copy, scripts/train.py:65-68).

Data parallel: start it under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
scripts/train.py ...` -- one process per GPU, frames sharded over the ranks, model replicated (rank 0's initialisation is
broadcast), gradients averaged by the RCCL all-reduce inside the fused step, checkpoints written by rank 0 only.
`--batch-size` is the per-GPU batch; `--global-batch B` fixes the global batch instead (B / N rays per GPU)."""
import math
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import autolabel_amd  # noqa: F401  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before HIP initialises)
import torch
from torch import optim

from autolabel_amd import model_utils
from autolabel_amd import parallel
from autolabel_amd.dataset import ArrayDataset, LenDataset, SceneDataset
from autolabel_amd.trainer import SimpleTrainer


def read_args():
    parser = model_utils.model_flag_parser()
    parser.add_argument('scene')
    parser.add_argument('--factor-train', type=float, default=2.0)
    parser.add_argument('--factor-test', type=float, default=2.0)
    parser.add_argument('--batch-size', '-b', type=int, default=4096)
    parser.add_argument('--iters', type=int, default=10000)
    parser.add_argument('--workers', '-w', type=int, default=1)
    parser.add_argument('--eval', action='store_true')
    parser.add_argument('--workspace', type=str, default=None, help='Save results here instead of the scene directory.')
    parser.add_argument('--synthetic', choices=['room', 'cube'], default=None)
    parser.add_argument('--device-data', action='store_true', help='(default whenever the frames fit in HBM; kept for older command lines)')
    parser.add_argument('--host-data', action='store_true', help="the reference's host DataLoader instead of device-resident frames")
    parser.add_argument('--shard-optimizer', action='store_true', help='data parallel: every rank runs Adam on 1 / world of the hash table (reduce-scatter + all-gather)')
    parser.add_argument('--dp-level-group', type=int, default=4, help='data parallel: hash-grid levels per gradient bucket of the overlapped exchange '
                        '(4 = five buckets, 8 = three: less scatter time, a larger last bucket with nothing to hide behind)')
    parser.add_argument('--dp-graph', action='store_true', help='data parallel: capture the step with its RCCL collectives into a hipGraph')
    parser.add_argument('--global-batch', type=int, default=0, help='data parallel: fixed global batch (overrides --batch-size)')
    parser.add_argument('--cuda-ray', action='store_true', help='occupancy-grid marching (the reference hard-codes cuda_ray=False)')
    parser.add_argument('--march-samples', type=int, default=96, help='sample rows per ray with --cuda-ray')
    return parser.parse_args()


def main():
    flags = read_args()
    rank, world, local = parallel.init_distributed()     # before any GPU call; a single process is rank 0 of 1
    device = f'cuda:{local}'
    if world > 1:
        torch.cuda.set_device(local)
        np.random.seed(parallel.rank_seed(0, rank) % (2 ** 32))   # host-side frame / pixel picks differ per rank
        if flags.global_batch:
            assert flags.global_batch % (512 * world) == 0, 'global batch must be a multiple of 512 rays per rank'
            flags.batch_size = flags.global_batch // world
    if flags.synthetic:
        from autolabel_amd import synthetic
        scene = synthetic.make_room_scene(n_frames=60, feat_dim=64 if flags.features else 0) if flags.synthetic == 'room' \
            else synthetic.make_cube_scene()
        dataset = ArrayDataset(synthetic.subsample(scene, flags.factor_train), batch_size=flags.batch_size)
    else:
        dataset = SceneDataset('train', flags.scene, factor=flags.factor_train, batch_size=flags.batch_size, features=flags.features)
    n_classes = dataset.n_classes if dataset.n_classes is not None else 2
    kw = dict(cuda_ray=True, march_samples=flags.march_samples) if flags.cuda_ray else {}
    model = model_utils.create_model(dataset.min_bounds, dataset.max_bounds, n_classes, flags, **kw)
    opt = Namespace(rand_pose=-1, color_space='srgb', feature_loss=flags.features is not None, rgb_weight=flags.rgb_weight,
                    depth_weight=flags.depth_weight, semantic_weight=flags.semantic_weight, feature_weight=flags.feature_weight)
    optimizer = lambda model: torch.optim.Adam([
        {'name': 'encoding', 'params': list(model.encoder.parameters())},
        {'name': 'net', 'params': model.network_parameters(), 'weight_decay': 1e-6},
    ], lr=flags.lr, betas=(0.9, 0.99), eps=1e-15)
    # the reference's loader; SimpleTrainer replaces it by device-resident frames when they fit (device_data='auto')
    train_dataloader = torch.utils.data.DataLoader(LenDataset(dataset, 1000), batch_size=None, num_workers=flags.workers)
    train_dataloader._data = dataset
    criterion = torch.nn.MSELoss(reduction='none')
    gamma = 0.5
    steps = math.log(1e-4 / flags.lr, gamma)
    step_size = max(flags.iters // steps // 1000, 1)
    scheduler = lambda optimizer: optim.lr_scheduler.StepLR(optimizer, gamma=gamma, step_size=step_size)
    epochs = int(np.ceil(flags.iters / 1000))
    model_dir = model_utils.model_dir(flags.scene, flags)
    if rank == 0:
        model_utils.write_params(model_dir, flags)
    pg = torch.distributed.group.WORLD if world > 1 else None
    trainer = SimpleTrainer('ngp', opt, model, device=device, workspace=model_dir, optimizer=optimizer, criterion=criterion,
                            fp16=True, ema_decay=0.95, lr_scheduler=scheduler, scheduler_update_every_step=False, metrics=[],
                            use_checkpoint='latest', local_rank=rank, world_size=world, process_group=pg,
                            device_data=False if flags.host_data else ('auto' if not flags.device_data else True), use_graph_dp=flags.dp_graph,
                            shard_optimizer=flags.shard_optimizer, dp_level_group=flags.dp_level_group)
    if world > 1:   # replicas start identical (a resumed checkpoint is read by every rank; the broadcast also covers a fresh init)
        parallel.broadcast_parameters(model._ensure_device().P.flat, pg)
        model._shadow_version = None
    trainer.train(train_dataloader, epochs)
    if rank == 0 or trainer.shard_optimizer:   # (sharded optimizer: gathering the state is a collective, rank 0 writes)
        trainer.save_checkpoint()
    if world > 1:
        torch.distributed.barrier()
    if flags.eval and rank == 0:
        if flags.synthetic:
            testset = ArrayDataset(synthetic.subsample(scene, flags.factor_test), batch_size=flags.batch_size * 2, split='test')
        else:
            testset = SceneDataset('test', flags.scene, factor=flags.factor_test, batch_size=flags.batch_size * 2)
        test_dataloader = torch.utils.data.DataLoader(LenDataset(testset, testset.rotations.shape[0]), batch_size=None, num_workers=0)
        trainer.evaluate(test_dataloader)


if __name__ == '__main__':
    main()
