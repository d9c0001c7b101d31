import sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
class A: pass
a = A(); a.frames = 200; a.feature_dim = 64; a.batch = 4096
dev = torch.device('cuda', 0)
scene, full, train, eng, fr = bench.build(a, dev, 0, 1)
batch = train.alloc_batch(4096)
def step(i):
    train.next_train(batch, seed=1234, step=i, frame_range=fr)
    eng.step(batch, seed=99, step=i)
for i in range(20): step(i)
torch.cuda.synchronize()
t0 = time.time()
for i in range(100): step(20 + i)
t_cpu = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
print(f'CPU issue time {t_cpu*10:.2f} ms/step ; wall {t_all*10:.2f} ms/step')
