"""autolabel_amd -- MI355X-native NeRF train/render core for ethz-asl/autolabel's hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of
the hot path runs in hand-written HIP kernels (csrc/) behind the C ABI declared in
include/autolabel_hip.h.  There is no CPU fallback: the compute entry points raise if the HIP
library or a GPU is missing.
"""
__version__ = '0.1.0'
