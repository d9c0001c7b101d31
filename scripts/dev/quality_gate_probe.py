"""Dev probe: tests/test_gpu_quality.run_gate under other regimes (steps, scene size, grid size, seeds) -> one JSON line each.
usage: python scripts/dev/quality_gate_probe.py steps=1500 w=96 h=72 ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import test_gpu_quality as Q

cfg = dict(Q.GATE)
for a in sys.argv[1:]:
    k, v = a.split('=')
    cfg[k] = tuple(int(x) for x in v.split(',')) if k.endswith('seeds') else (v if k in ('oracle_device',) else (v == 'True' if v in ('True', 'False') else int(v)))
t0 = time.time()
rec = Q.run_gate(**cfg)
rec['wall_s'] = time.time() - t0
rec['cfg'] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
print(json.dumps({k: v for k, v in rec.items() if not k.startswith('loss_')}), flush=True)
