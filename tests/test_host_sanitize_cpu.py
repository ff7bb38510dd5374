"""Host code of librt355 (validation, the four tree builders on several threads, flattening,
Sobol' tables) and the whole CPU oracle under AddressSanitizer + UBSan, and again under ThreadSanitizer.  The build compiles the
kernels once more (about two minutes), so the test only runs when RT355_SANITIZE=1."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("RT355_SANITIZE") != "1", reason="set RT355_SANITIZE=1 (slow: rebuilds the library)")
@pytest.mark.parametrize("mode", ["address", "thread"])
def test_host_code_under_sanitizers(tmp_path, mode):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "tests", "sanitize", "run.sh"), mode], env=env, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0 and "host_check ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
