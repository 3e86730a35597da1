"""Every variant library that tools/build_variants.py left under miniengineao_amd/lib/variants/ (experimental -D arms
of meao_kernels.hip, A/B candidates) must pass the same parity smoke as the product: an arm that stays in the
source cannot rot unseen.  No variants built -> nothing to check."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = sorted(glob.glob(os.path.join(ROOT, "miniengineao_amd", "lib", "variants", "libmeao_*.so")))


@pytest.mark.gpu
@pytest.mark.parametrize("lib", VARIANTS or [None], ids=lambda p: os.path.basename(p) if p else "none-built")
def test_variant_library_is_bit_exact(lib):
    if lib is None:
        pytest.skip("no variant libraries under miniengineao_amd/lib/variants/")
    env = dict(os.environ, MEAO_LIB_PATH=lib)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_smoke.py")], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-1500:])
