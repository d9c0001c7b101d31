import os as _os

# ROCm 7.2's hipGraph "packet capture" fast path (AQL packets and kernel arguments pre-built at instantiate) is corrupted by
# ordinary kernel launches issued between two launches of the graph: the next replay then runs with stale kernel arguments
# (NaNs, "Memory access fault ... write access to a read-only page").  Reproducer: scripts/dev/debug_graph_eager.py.  The flag
# must be in the environment before the HIP runtime initialises (first GPU call), so it is set on import; replay speed of the
# training step is unchanged (2.26 ms at B = 4096 either way).
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

"""autolabel_amd -- MI355X-native NeRF train/render core for ethz-asl/autolabel's hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of
the hot path runs in hand-written HIP kernels (csrc/) behind the C ABI declared in
include/autolabel_hip.h.  There is no CPU fallback: the compute entry points raise if the HIP
library or a GPU is missing.
"""
__version__ = '0.1.0'
