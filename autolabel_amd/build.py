"""Build libautolabel_hip.so for gfx950 with hipcc (in-tree, so it travels with the repo snapshot).

Every translation unit is compiled to its own object (in parallel, and only when it or a header changed), then linked."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libautolabel_hip.so')
OBJ = os.path.join(CSRC, 'build')
SOURCES = ['encode.hip', 'mlp.hip', 'mlp_bwd128.hip', 'mlp_fwd128.hip', 'sampling.hip', 'heads.hip', 'raygen.hip', 'loss.hip', 'adam.hip', 'march.hip', 'wide.hip',
           'capi.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-Wno-unused-value', '-fPIC',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']  # MFMA results straight into VGPRs (no v_accvgpr_read for every epilogue)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'autolabel_hip.h'))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return _stale(LIB, srcs + _headers())


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    heads = _headers()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, s + '.o')
        jobs.append((src, obj, force or _stale(obj, [src] + heads)))

    def compile_one(job):
        src, obj, stale = job
        if stale:
            cmd = ['hipcc'] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print('[autolabel_amd] ' + ' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, jobs))
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print('[autolabel_amd] ' + ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build_library(force=True)
