"""Seeded option/data fuzz of the HIP path against the oracle and the real reference decoder
(tools/gpu_fuzz.py): presets, span sizes, custom depths / nice_len / dictionary / lc-lp-pb, BCJ, checks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_against_oracle(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), str(seed), "40"],
                       capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert f"FUZZ seed {seed}: 40 / 40 ok" in p.stdout
