"""The matched-quality gate of tests/test_gpu_quality.py with other seed counts / model sizes: more seeds per side
than the in-suite run can afford, and one seed pair at the bench's own model (L = 16, T = 2^19, 128 + 128 samples).
usage: quality_gate_run.py <out.json> [hip_seeds=7,8,9,10,11] [oracle_seeds=7,8,9,10,11] [levels=12] [log2_T=17] [s1=48] [s2=48] [steps=4000] [batch=1024] [w=96] [h=72] [n_frames=40]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import autolabel_amd  # noqa
import test_gpu_quality as Q
cfg = dict(Q.GATE)
for a in sys.argv[2:]:
    k, v = a.split('=')
    cfg[k] = tuple(int(x) for x in v.split(',')) if k.endswith('seeds') else int(v)
rec = Q.run_gate(**cfg)
rec['config'] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
json.dump(rec, open(sys.argv[1], 'w'), indent=1)
print(json.dumps({k: v for k, v in rec.items() if not k.startswith('loss_') and k not in ('hip', 'oracle_runs')}))
