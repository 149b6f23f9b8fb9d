import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """Leave the GPU in a quiet state before the interpreter (and torch's / HIP's static objects) are torn down: destroy
    every library handle that is still alive -- streams, graphs, events -- and drain the device."""
    net = sys.modules.get("mobileposer_amd.net")
    if net is not None and hasattr(net, "_close_all"):
        net._close_all()
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
    import gc
    gc.collect()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def weights():
    from mobileposer_amd.synthetic import make_weights
    return make_weights(0)


@pytest.fixture(scope="session")
def smpl():
    from mobileposer_amd.synthetic import synthetic_smpl
    return synthetic_smpl()


def geodesic(Ra, Rb):
    """Angle (rad) between rotation matrices [...,3,3] (float64 inside)."""
    Ra = np.asarray(Ra, dtype=np.float64)
    Rb = np.asarray(Rb, dtype=np.float64)
    D = np.swapaxes(Ra, -1, -2) @ Rb
    # robust: angle = 2*asin(||D - I||_F / (2*sqrt(2)))
    n = np.linalg.norm(D - np.eye(3), axis=(-1, -2))
    return 2.0 * np.arcsin(np.clip(n / (2.0 * np.sqrt(2.0)), 0.0, 1.0))


# ---- shared by the -m gpu test modules -----------------------------------------------------------
@pytest.fixture(scope="session")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch


def lstm_test_modes():
    """Operand modes the shared fixtures run.  Default: exact-fp32 MFMA operands only (LSTM mode 1, the library default and the
    reference's arithmetic).  Rounds 2-5 ran every parity test a second time in the opt-in split-fp16 mode 3; round 6 added the
    lo*lo product to it and measured it again at 256 x 125 on trained-regime weights: 1.8-3.5 x the fp32 oracle's distance from
    float64, above the 2.0 x the exact mode is held to (profiles/r06_mode3.txt) -- it is not a second fp32, so it no longer rides
    through the whole suite.  What is left of it: test_mode3_smoke (goldens G2 / G5), the 256 x 125 accuracy record
    (tests/test_gpu_round4.py, held to 5 x) and the bitwise x3 / x3w cross-check.  MP_TEST_MODES=fp32,x3 brings the old suite back."""
    return [m for m in os.environ.get("MP_TEST_MODES", "fp32").split(",") if m in ("fp32", "x3")] or ["fp32"]


@pytest.fixture(params=lstm_test_modes())
def net(request, torch_mod, weights, smpl):
    """A handle per test in every mode of lstm_test_modes() -- same goldens, same oracle, same tolerances.
    The handle is closed when the test ends: one live native handle at a time unless a test builds more itself."""
    from mobileposer_amd.net import MobilePoserNet
    n = MobilePoserNet.from_numpy(weights, smpl, device="cuda:0")
    n.lstm_mode = 3 if request.param == "x3" else 1
    n.set_lstm_mode(n.lstm_mode)
    yield n
    n.close()


def cu(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def npy(t):
    return t.detach().cpu().numpy()
