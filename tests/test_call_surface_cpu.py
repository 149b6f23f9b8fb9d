"""The call surface of the path against the reference's (golden G21: parameter names and literal defaults of the reference's
callables as inspect.signature reports them, recorded by tests/golden/make_golden.py -- SURVEY.md 8(b)).  A caller written against
the reference must be able to make the same call here: every reference parameter is there, in the same position, under the
same name and with the same literal default; what this package adds behind them has a default of its own.  CPU only (classes
and functions, no instance) -- the attributes of an instance are checked in tests/test_gpu_round6.py."""
import inspect
import json
import os

import pytest

from conftest import GOLDEN


def _surface():
    with open(os.path.join(GOLDEN, "g21_call_surface.json")) as f:
        return json.load(f)


def _ours():
    from mobileposer_amd import evaluate, net
    from mobileposer_amd.body_model import ParametricModel
    from mobileposer_amd.data import PoseDataset
    from mobileposer_amd.model_utils import load_model
    return {"MobilePoserNet": net.MobilePoserNet, "Joints": net._ModuleView, "Poser": net._PoserView, "FootContact": net._ModuleView,
            "Velocity": net._VelocityView, "ParametricModel": ParametricModel, "PoseDataset": PoseDataset,
            "PoseEvaluator": evaluate.PoseEvaluator,
            "functions": type("F", (), {"load_model": staticmethod(load_model), "evaluate_pose": staticmethod(evaluate.evaluate_pose)})}


def _params(fn):
    out = []
    for name, prm in inspect.signature(fn).parameters.items():
        if name in ("self", "cls"):
            continue
        out.append((name, prm.default, prm.kind))
    return out


CASES = [(owner, name) for owner, fns in _surface().items() if isinstance(fns, dict) for name in fns]


@pytest.mark.parametrize("owner,name", CASES)
def test_reference_call_signatures_are_accepted(owner, name):
    ref = _surface()[owner][name]
    fn = inspect.getattr_static(_ours()[owner], name)
    fn = fn.__func__ if isinstance(fn, (staticmethod, classmethod)) else fn
    mine = _params(fn)
    assert len(mine) >= len(ref), (owner, name, [m[0] for m in mine], [r[0] for r in ref])
    for (rname, rdefault), (mname, mdefault, kind) in zip(ref, mine):
        assert kind in (inspect.Parameter.POSITIONAL_OR_KEYWORD,), (owner, name, mname)
        assert mname == rname, (owner, name, mname, rname)
        if rdefault is None:                                   # required in the reference: required, or optional, here
            continue
        assert mdefault is not inspect.Parameter.empty, (owner, name, mname)
        if rdefault != "<object>":                             # (an object default, e.g. torch.device('cpu'): this package runs on cuda)
            assert repr(mdefault) == rdefault, (owner, name, mname, mdefault, rdefault)
    for mname, mdefault, kind in mine[len(ref):]:              # what this package adds can be left out
        assert mdefault is not inspect.Parameter.empty or kind in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL), (owner, name, mname)


def test_surface_fixture_is_what_the_survey_names():
    s = _surface()
    assert s["MobilePoserNet_submodules"] == ["foot_contact", "joints", "pose", "velocity"]
    for a in ("num_past_frames", "num_future_frames", "num_total_frames", "floor_y", "feet_pos", "last_root_pos", "current_root_y", "imu"):
        assert a in s["MobilePoserNet_attributes"], a          # SURVEY.md 8(b), "Signatures"
