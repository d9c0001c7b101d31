"""The shipped library must not contain packed fp32 instructions.

Round 5 traced the red data-parallel lock-step test of round 4 (two ranks as two processes on ONE GPU) to the hardware, not to a
race in the kernels: on the MI355X boxes of this build a v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose op_sel picks the high
register of a freshly written pair returns +0 in lanes 48..63 now and then while another wave on the same SIMD executes
v_mfma_f32_32x32x16_f16 (scripts/dev/probe_pk_f32.hip reproduces it in isolation; profiles/r05_pk_f32_probe.txt).  hipcc emits those
forms for ordinary float2-shaped arithmetic (the interpolation weights and weight x gradient products of the binned scatter), so
autolabel_amd/build.py switches the `packed-fp32-ops` target feature off for every translation unit.  This test disassembles the
device code of the built library and fails if a later change (a new flag set, hand-written assembly) brings them back."""
import os
import shutil
import subprocess
import tempfile

import pytest

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PACKED_F32 = ('v_pk_mul_f32', 'v_pk_fma_f32', 'v_pk_add_f32')


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_no_packed_fp32_instructions_in_the_shipped_library():
    from autolabel_amd import build
    assert os.path.exists(build.LIB), 'build the library first (python -m autolabel_amd.build)'
    assert '-packed-fp32-ops' in build.FLAGS
    tmp = tempfile.mkdtemp(prefix='aln_objdump_')
    try:
        lib = os.path.join(tmp, 'lib.so')
        shutil.copy(build.LIB, lib)
        subprocess.run([OBJDUMP, '--offloading', lib], cwd=tmp, check=True, capture_output=True)    # extracts one code object per translation unit
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if 'amdgcn' in f and os.path.getsize(os.path.join(tmp, f))]
        assert len(objs) >= 10, f'expected one gfx950 code object per .hip source, found {len(objs)}'
        found, mfma = {}, 0
        for o in objs:
            asm = subprocess.run([OBJDUMP, '-d', o], check=True, capture_output=True, text=True).stdout
            mfma += asm.count('v_mfma_f32_32x32x16_f16')
            for op in PACKED_F32:
                n = asm.count(op)
                if n:
                    found[(os.path.basename(o), op)] = n
        assert mfma > 100, 'the disassembly does not look like this library (no MFMA instructions found)'
        assert not found, f'packed fp32 instructions in the shipped device code: {found}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
