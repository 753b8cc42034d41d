"""CPU-only: the numpy prototypes that pin the index algebra of the register / LDS transforms
(tools/proto_*.py: lane layouts, exchange images, real-input splits, the cepstrogram wave kernels'
row / even-extension / lifter / 4096-combine logic) still hold -- each script asserts against
numpy.fft or the restatement of the reference and exits non-zero on a mismatch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["proto_fft1024.py", "proto_fft1024_v2.py", "proto_fft512.py", "proto_cepstrogram_wave.py",
                                    "proto_cqt_f16.py", "proto_gemm_bf16.py", "proto_cqt_pyramid.py", "proto_fft256.py"])
def test_prototype_script(script):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "OK" in res.stdout
