"""30 launch-by-launch training steps of the LSeg configuration (D = 512, semantic_weight = 0) for a kernel profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse, torch
import bench
sys.argv = [sys.argv[0], "--feature-dim", "512"]; args = bench.parse()
args.feature_dim = 512
wl = bench.Workload(args, torch.device('cuda', 0))
eng = wl.engine(bench.SEEDS[0][0], feature_dim=512, semantic_weight=0.0)
batch = wl.train.alloc_batch(4096)
for it in range(30):
    wl.train.next_train(batch, seed=1, step=it)
    eng.step(batch, seed=2, step=it)
torch.cuda.synchronize()
print('done', eng.terms.tolist())
