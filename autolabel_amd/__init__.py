"""autolabel_amd -- MI355X-native NeRF train/render core for ethz-asl/autolabel's hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of
the hot path runs in hand-written HIP kernels (csrc/) behind the C ABI declared in
include/autolabel_hip.h.  There is no CPU fallback: the compute entry points raise if the HIP
library or a GPU is missing.
"""
import os as _os

# ROCm 7.2's hipGraph "packet capture" fast path (AQL packets and kernel arguments pre-built at instantiate) is corrupted by
# ordinary kernel launches issued between two launches of the graph: the next replay then runs with stale kernel arguments
# (NaNs, "Memory access fault ... write access to a read-only page").  Reproducer: scripts/dev/debug_graph_eager.py.  The flag
# must be in the environment before the HIP runtime initialises (first GPU call), so it is set on import; replay speed of the
# training step is unchanged (2.26 ms at B = 4096 either way).  Whether that worked is recorded: if the host program had
# already touched the GPU, or exported the variable as something else, graph_replay_is_safe() says no and the trainers issue
# their steps launch by launch instead (engine.GraphedStep refuses to capture).
_PACKET_VAR = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'
_preset = _os.environ.get(_PACKET_VAR)
_hip_was_up = False
# a profiler's preloaded tool library (rocprofv3 --pmc ...) brings the HIP runtime up before this process runs a line of Python:
# torch.cuda.is_initialized() cannot see that, so an unset variable under a profiler counts as "too late"
_profiler_preloaded = any('rocprof' in _os.environ.get(v, '').lower() for v in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'))
if _preset is None:
    try:
        import torch as _torch
        _hip_was_up = bool(_torch.cuda.is_initialized())
    except Exception:   # pragma: no cover
        _hip_was_up = False
    _hip_was_up = _hip_was_up or _profiler_preloaded
    _os.environ[_PACKET_VAR] = '0'


def graph_replay_is_safe():
    """(ok, reason): hipGraph replays may be interleaved with ordinary launches only with packet capture off, and the switch
    only counts if it was in the environment before HIP initialised."""
    if _os.environ.get(_PACKET_VAR) != '0':
        return False, f'{_PACKET_VAR}={_os.environ.get(_PACKET_VAR)!r} (must be 0: ROCm 7.2 replays graphs with stale kernel arguments otherwise)'
    if _preset is None and _hip_was_up:
        return False, (f'the HIP runtime was initialised before autolabel_amd could set {_PACKET_VAR}=0; export it before the '
                       'first GPU call (or import autolabel_amd first)')
    return True, ''


__version__ = '0.2.0'
