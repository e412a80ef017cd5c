"""The kernel-level and solver-level parity tests once more in a child process whose device allocations start as all-ones
(ITS_DEBUG_POISON_MALLOC=1) and whose compute entry points find every CU's LDS full of a NaN pattern
(ITS_DEBUG_POISON_LDS=1): a kernel that reads memory or LDS it has not written then fails its parity test instead of
passing or failing by what happened to run before it (how the line sweep's uninitialised first window was found)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suites_pass_with_poisoned_allocations_and_lds():
    env = dict(os.environ, ITS_DEBUG_POISON_MALLOC="1", ITS_DEBUG_POISON_LDS="1")
    files = ["tests/test_gpu_kernels.py", "tests/test_gpu_solvers.py", "tests/test_gpu_sweep_lines.py", "tests/test_gpu_gs_prefix.py",
             "tests/test_gpu_ell.py", "tests/test_gpu_ellg.py", "tests/test_gpu_panel.py", "tests/test_gpu_mixed_precision.py"]
    files = [f for f in files if os.path.exists(os.path.join(ROOT, f))]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"] + files,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
