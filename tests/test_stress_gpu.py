"""GPU: a short, fixed-seed run of the randomised sweeps (tests/stress_parity.py, tests/stress_more.py, tests/stress_refine.py; run them for
minutes with other seeds when kernels change)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,seed,extra", [
    ("stress_parity.py", 5, []), ("stress_more.py", 6, []), ("stress_refine.py", 7, []),
    # the adversarial orientation fields (near-ties, perpendicular pairs, zero vectors, fans on the flat top of the cosine) with
    # the key body forced, and random 8-bit code maps through the key body (contexts of 8-bit views take the select body by
    # default): the key body's corner cases, every run
    ("stress_parity.py", 8, ["--ori-mode", "mix", "--body", "1"]), ("stress_parity.py", 9, ["--codes", "--body", "1"])])
def test_short_sweep(script, seed, extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), "--minutes", "0.2", "--seed", str(seed)] + extra,
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "'mismatching_cases': 0" in r.stdout
