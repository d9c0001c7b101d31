"""Micro-benchmark of the wide-head GEMMs on the LSeg shapes (1M rows): TFLOP/s of aln_wide_nt / aln_wide_tn."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import autolabel_amd, torch
from autolabel_amd import hip as H
M = 1 << 20
dev = 'cuda'
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
x = (torch.randn(M, 512, device=dev) * 0.5).half()
sout = (torch.randn(M, 16, device=dev) * 0.5).half()
y = torch.empty(M, 512, dtype=torch.float16, device=dev)
for (N, K1, geo, relu1, name) in [(512, 512, False, 0, 'plain 512x512'), (512, 0, True, 0, 'geo -> 512'), (64, 512, True, 1, 'relu(f)+geo -> 64'), (16, 512, False, 0, '512 -> 16')]:
    K = K1 + (16 if geo else 0)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    yy = y[:, :N].contiguous() if N != 512 else y
    t = timeit(lambda: H.call('aln_wide_nt', H.ptr(x) if K1 else None, 512, K1, relu1, H.ptr(sout) if geo else None, 15, M, N, H.ptr(w), K,
                              H.ptr(yy), N, 1, None, 0, None, 0, None, H.stream()))
    print(f'nt {name:22s}: {t * 1e6:8.0f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s')
mk = torch.randn(M, 512, device=dev).half()
t = timeit(lambda: H.call('aln_wide_nt', H.ptr(x), 512, 512, 0, None, 15, M, 512, H.ptr(w := (torch.randn(512, 512, device=dev) / 22).half()), 512,
                          H.ptr(y), 512, 0, H.ptr(mk), 512, H.ptr(y), 512, None, H.stream()))
print(f'nt mask+add 512x512       : {t * 1e6:8.0f} us  {2.0 * M * 512 * 512 / t / 1e12:7.1f} TFLOP/s')
g = (torch.randn(M, 512, device=dev) * 0.1).half()
dw = torch.zeros(512, 528, device=dev)
for (N, K1, geo, name) in [(512, 512, False, 'dW 512x512'), (64, 512, True, 'dW 64x528'), (512, 0, True, 'dW 512x16')]:
    K = K1 + (16 if geo else 0)
    gg = g[:, :N].contiguous() if N != 512 else g
    tn_ws = torch.empty(int(H.lib().aln_wide_tn_ws_bytes(M, N, K)), dtype=torch.uint8, device='cuda')
    t = timeit(lambda: H.call('aln_wide_tn', H.ptr(gg), N, H.ptr(x) if K1 else None, 512, K1, 0, H.ptr(sout) if geo else None, 15, M, N, H.ptr(dw), K, H.ptr(tn_ws), H.stream()))
    print(f'tn {name:22s}: {t * 1e6:8.0f} us  {2.0 * M * N * K / t / 1e12:7.1f} TFLOP/s')
# ---- round 5: generated first layer (h1 never stored)
w0 = (torch.randn(512, 16, device=dev) / 4).half()
w1 = (torch.randn(512, 512, device=dev) / 22).half()
t = timeit(lambda: H.call('aln_wide_nt_gen', H.ptr(sout), 15, H.ptr(w0), M, 512, 512, H.ptr(w1), 512, H.ptr(y), 512, 1, None, None, None, H.stream()))
print(f'nt_gen (layers 1+2)       : {t * 1e6:8.0f} us  {2.0 * M * 512 * 528 / t / 1e12:7.1f} TFLOP/s')
y2 = torch.empty_like(y)
t = timeit(lambda: H.call('aln_wide_nt_maskgen', H.ptr(x), 512, M, 512, 512, H.ptr(w1), 512, H.ptr(y2), 512, H.ptr(sout), 15, H.ptr(w0), None, H.stream()))
print(f'nt_maskgen 512x512        : {t * 1e6:8.0f} us  {2.0 * M * 512 * 512 / t / 1e12:7.1f} TFLOP/s')
t = timeit(lambda: H.call('aln_wide_nt', H.ptr(x), 512, 512, 0, None, 15, M, 512, H.ptr(w1), 512, H.ptr(y2), 512, 0, H.ptr(mk), 512, None, 0, None, H.stream()))
print(f'nt stored mask 512x512    : {t * 1e6:8.0f} us  {2.0 * M * 512 * 512 / t / 1e12:7.1f} TFLOP/s')
tn_ws = torch.empty(int(H.lib().aln_wide_tn_ws_bytes(M, 512, 512)), dtype=torch.uint8, device='cuda')
dw2 = torch.zeros(512, 512, device=dev)
t = timeit(lambda: H.call('aln_wide_tn_gen', H.ptr(g), 512, H.ptr(sout), 15, H.ptr(w0), M, 512, 512, H.ptr(dw2), 512, H.ptr(tn_ws), H.stream()))
print(f'tn_gen dW 512x512         : {t * 1e6:8.0f} us  {2.0 * M * 512 * 512 / t / 1e12:7.1f} TFLOP/s')
