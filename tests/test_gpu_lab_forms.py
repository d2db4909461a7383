"""The forced launch forms of the GEMM live in the LAB build only (tools/lab/, -DP2T_LAB): their fuzz / edge-shape cases
(tests/lab_forms_cases.py) run here in a child process that loads tools/build/libp2t_lab.so through P2T_HIP_LIB.  The product
library of this process is untouched (and refuses those policies: tests/test_gpu_kernels.py::test_gemm_argument_errors)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "build", "libp2t_lab.so")


@pytest.mark.gpu
def test_forced_launch_forms_on_the_lab_build():
    assert os.path.exists(LAB), f"{LAB} missing: __graft_entry__.build() / `make -C prot2text-v2-esm3_amd/csrc lab` builds it"
    env = dict(os.environ, P2T_HIP_LIB=LAB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "lab_forms_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-4000:], r.stderr[-2000:])
    assert " passed" in r.stdout
