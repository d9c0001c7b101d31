"""aln_lzf_decompress (host code of the C-ABI library) under AddressSanitizer + UBSan on the CPU: 20 000 valid, truncated,
under-sized and damaged LZF streams against exactly-sized heap buffers (tests/native/lzf_fuzz.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_lzf_decoder_is_memory_safe_under_asan(tmp_path):
    exe = tmp_path / 'lzf_fuzz'
    cmd = ['g++', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-std=c++17',
           os.path.join(ROOT, 'autolabel_amd', 'csrc', 'capi.cpp'), os.path.join(ROOT, 'tests', 'native', 'lzf_fuzz.cpp'), '-o', str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0'))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'round trips' in r.stdout
