"""Build libautolabel_hip.so for gfx950 with hipcc (in-tree, so it travels with the repo snapshot)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libautolabel_hip.so')
SOURCES = ['encode.hip', 'mlp.hip', 'sampling.hip', 'heads.hip', 'raygen.hip', 'loss.hip', 'adam.hip', 'march.hip', 'wide.hip',
           'capi.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-Wno-unused-value', '-fPIC', '-shared',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1']  # MFMA results straight into VGPRs (no v_accvgpr_read for every epilogue)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.h'))]
    deps.append(os.path.join(os.path.dirname(CSRC), '..', 'include', 'autolabel_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ['hipcc'] + FLAGS + srcs + ['-o', LIB + '.tmp']
    if verbose:
        print('[autolabel_amd] ' + ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build_library(force=True)
