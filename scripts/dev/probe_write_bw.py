"""Dev only: what a plain streaming write / copy of the record pool's size costs on this box (torch kernels: fill, copy)."""
import torch
n = 743 * 2 ** 20
a = torch.empty(n, dtype=torch.uint8, device='cuda'); b = torch.empty(n, dtype=torch.uint8, device='cuda')
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
af, bf = a.view(torch.float32), b.view(torch.float32)
us = t(lambda: af.zero_()); print(f'zero_ 743 MB: {us:.1f} us = {n / us / 1e6:.2f} TB/s written')
us = t(lambda: bf.copy_(af)); print(f'copy_ 743 MB: {us:.1f} us = {n / us / 1e6:.2f} TB/s read + the same written')
x = torch.empty(2 * n, dtype=torch.uint8, device='cuda').view(torch.float32)
us = t(lambda: x.zero_()); print(f'zero_ 1486 MB: {us:.1f} us = {2 * n / us / 1e6:.2f} TB/s written')
s = torch.empty(64 * 2 ** 20, dtype=torch.uint8, device='cuda').view(torch.float32)
us = t(lambda: s.zero_()); print(f'zero_ 64 MB (fits the memory-side cache): {us:.1f} us = {64 * 2**20 / us / 1e6:.2f} TB/s')
