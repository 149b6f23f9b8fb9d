"""The C ABI used from plain C (examples/cabi_smoke.c): built with gcc against include/mobileposer_hip.h and the
in-tree shared library, run on the GPU -- no Python, no torch in that process."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_plain_c_consumer_of_the_abi(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc / HIP headers not available")
    libdir = os.path.join(REPO, "mobileposer_amd")
    exe = str(tmp_path / "cabi_smoke")
    subprocess.check_call([gcc, "-O2", "-std=c11", os.path.join(REPO, "examples", "cabi_smoke.c"),
                           "-I" + os.path.join(REPO, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           "-L" + libdir, "-lmobileposer_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_smoke: ok" in r.stdout
