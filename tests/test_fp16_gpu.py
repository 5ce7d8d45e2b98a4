"""The fp16 build (libdanhip_f16.so) is selected per process by DANHIP_DTYPE=fp16, so its cases run in one child process
(tests/fp16/cases.py): conv kernel families, HBM-bound layers, S3FD forward parity against the oracle in fp16-storage
emulation, DAN-Deform training steps with the static loss scale."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp16_build_cases(dev):
    env = dict(os.environ, DANHIP_DTYPE="fp16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fp16", "cases.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FP16-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
