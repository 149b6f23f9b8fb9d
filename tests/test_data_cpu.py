"""PoseDataset eval-mode input formation against the reference's own PoseDataset output (golden G8). CPU only."""
import numpy as np
import torch

from conftest import load_golden
from mobileposer_amd.data import PoseDataset
from oracle import mp_oracle as O


def test_g8_dataset_formation(smpl):
    g = load_golden("g8_dataset.npz")
    data = {k: [torch.from_numpy(g[f"in{i}_{k}"]) for i in range(2)] for k in ("acc", "ori", "pose", "tran")}

    def fk(pose):                       # test-side FK (oracle) standing in for the GPU kernel
        Rg, jg = O.forward_kinematics(pose.numpy(), smpl["J"])
        return torch.from_numpy(Rg), torch.from_numpy(jg)

    ds = PoseDataset(fold='test', evaluate='dip', data=data, fk=fk)
    assert len(ds) == int(g["n_items"]) == 24                      # 2 sequences x 12 combos
    for idx in range(len(ds)):
        imu, pose, joint, tran = ds[idx]
        assert np.array_equal(imu.numpy(), g[f"item{idx}_imu"])
        assert np.abs(pose.numpy() - g[f"item{idx}_pose"]).max() == 0
        assert np.abs(joint.numpy() - g[f"item{idx}_joint"]).max() < 1e-5
        assert np.array_equal(tran.numpy(), g[f"item{idx}_tran"])
    # combo masks: item 0 is 'lw_rp_h' = devices [0,3,4]; the other two devices are zero
    imu0 = ds[0][0]
    assert float(imu0[:, 3:9].abs().max()) == 0 and float(imu0[:, 0:3].abs().max()) > 0


def test_reference_constructor_reads_the_configured_file(smpl, tmp_path, monkeypatch):
    """PoseDataset(fold='test', evaluate='dip') as evaluate.py:124 calls it: resolves
    paths.processed_datasets/eval/dip_test.pt (config.py:33-34,104-108).  The ground-truth joints come from a callable
    here (the oracle, test-side): the default -- a GPU ParametricModel -- is exercised by tests/test_gpu_boundary.py."""
    from mobileposer_amd import config
    g = load_golden("g8_dataset.npz")
    data = {k: [torch.from_numpy(g[f"in{i}_{k}"]) for i in range(2)] for k in ("acc", "ori", "pose", "tran")}
    (tmp_path / "eval").mkdir()
    torch.save(data, tmp_path / "eval" / "dip_test.pt")
    monkeypatch.setattr(config.paths, "processed_datasets", tmp_path)

    def fk(pose):
        Rg, jg = O.forward_kinematics(pose.numpy(), smpl["J"])
        return torch.from_numpy(Rg), torch.from_numpy(jg)

    ds = PoseDataset(fold='test', evaluate='dip', fk=fk)
    assert len(ds) == 24
    for idx in (0, 5, 13, 23):
        imu, pose, joint, tran = ds[idx]
        assert np.array_equal(imu.numpy(), g[f"item{idx}_imu"])
        assert np.abs(pose.numpy() - g[f"item{idx}_pose"]).max() == 0
        assert np.abs(joint.numpy() - g[f"item{idx}_joint"]).max() < 1e-5
    import pytest
    with pytest.raises(ValueError):
        PoseDataset(fold='test', evaluate='nope', fk=fk)
    with pytest.raises(ValueError):
        PoseDataset(fold='dev', fk=fk)


class _Hyper:                         # stands for a Lightning hyper-parameter object: not a tensor, not a plain container
    def __init__(self):
        self.lr = 1e-3


def test_checkpoints_are_unpickled_in_full_only_when_trusted(tmp_path):
    """load_model / PoseDataset read files with torch's safe loader; a file that needs arbitrary unpickling is refused
    unless the caller says it is trusted (no silent fallback)."""
    import pytest
    from mobileposer_amd.model_utils import safe_torch_load
    plain, fancy = tmp_path / "plain.pth", tmp_path / "fancy.ckpt"
    torch.save({"state_dict": {"w": torch.ones(2)}, "epoch": 3, "hyper_parameters": {"finetune": False}}, plain)
    torch.save({"state_dict": {"w": torch.ones(2)}, "hyper_parameters": _Hyper()}, fancy)
    assert safe_torch_load(plain)["epoch"] == 3
    with pytest.raises(RuntimeError, match="trusted=True"):
        safe_torch_load(fancy)
    assert safe_torch_load(fancy, trusted=True)["hyper_parameters"].lr == 1e-3


def test_untrusted_dataset_file_is_not_swallowed_as_a_broken_file(tmp_path, monkeypatch):
    """The per-file ``except Exception: print(...)`` of the source loop (data.py:50-54) must not turn "this file needs
    trusted=True" into an empty dataset: the refusal escapes, and trusted=True reads the file."""
    import pytest
    from mobileposer_amd import config
    from mobileposer_amd.model_utils import UntrustedFileError
    g = load_golden("g8_dataset.npz")
    data = {k: [torch.from_numpy(g[f"in{i}_{k}"]) for i in range(2)] for k in ("acc", "ori", "pose", "tran")}
    data["meta"] = _Hyper()                                    # something the safe loader refuses
    (tmp_path / "eval").mkdir()
    torch.save(data, tmp_path / "eval" / "dip_test.pt")
    monkeypatch.setattr(config.paths, "processed_datasets", tmp_path)
    fk = lambda pose: (pose, torch.zeros(pose.shape[0], 24, 3))
    with pytest.raises(UntrustedFileError, match="trusted=True"):
        PoseDataset(fold='test', evaluate='dip', fk=fk)
    assert len(PoseDataset(fold='test', evaluate='dip', fk=fk, trusted=True)) == 24
