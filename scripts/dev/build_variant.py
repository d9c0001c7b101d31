"""Dev only: build a VARIANT of libautolabel_hip.so for a same-box A/B (the product has no switches).
usage: build_variant.py <tag> <source>[:-DFOO=1[,-DBAR=2]] ...   -> scripts/dev/_build/lib_<tag>.so
The named sources are recompiled with the extra defines, every other object is the product's (autolabel_amd/csrc/build)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autolabel_amd import build as B

tag = sys.argv[1]
out = os.path.join(ROOT, "scripts", "dev", "run", "variants")
os.makedirs(os.path.join(out, tag), exist_ok=True)
B.build_library(verbose=False)
objs = {s: os.path.join(B.OBJ, s + '.o') for s in B.SOURCES}
procs = []
for spec in sys.argv[2:]:
    src, _, defs = spec.partition(':')
    o = os.path.join(out, tag, src + '.o')
    procs.append(subprocess.Popen(['hipcc'] + B.FLAGS + [d for d in defs.split(',') if d] + ['-c', os.path.join(B.CSRC, src), '-o', o], stderr=subprocess.DEVNULL))
    objs[src] = o
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out, 'lib_%s.so' % tag)
subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + list(objs.values()) + ['-o', lib], check=True)
print(lib)
