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
         '-mllvm', '-amdgpu-mfma-vgpr-form=1',  # MFMA results straight into VGPRs (no v_accvgpr_read for every epilogue)
         # No packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  On the MI355X boxes of this build a v_pk_mul_f32
         # whose op_sel picks the HIGH register of a freshly written pair returns +0 in lanes 48..63 now and then while a second
         # process runs MFMA-heavy kernels on the same GPU (scripts/dev/probe_pk_f32.hip reproduces it in isolation; 1 product in
         # 3e7, every SIMD of the chip; profiles/r05_pk_f32_probe.txt).  That was the run-to-run difference behind the red
         # data-parallel lock-step test of round 4: two ranks on one GPU, interpolation weights and gradient products of the
         # binned scatter (encode.hip) zeroed for a quarter wave.  The scalar forms cost nothing measurable.
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
HOST_NOISE = "is not a recognized feature for this target"   # the host half of a .hip compile sees the device feature switch too


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
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            err = '\n'.join(l for l in r.stderr.splitlines() if HOST_NOISE not in l)
            if err.strip():
                print(err, flush=True)
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
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
