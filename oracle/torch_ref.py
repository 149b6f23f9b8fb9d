"""Second CPU baseline: the same path written with the torch building blocks the reference itself uses
(``nn.Linear`` / ``nn.LSTM`` / ``pack_padded_sequence``, models/rnn.py:13-33) so that the CPU number in bench.py
reflects ATen's (oneDNN / MKL) LSTM on the host cores rather than numpy.  TEST / BENCH INFRASTRUCTURE ONLY --
nothing under ``mobileposer_amd/`` imports this file.  Kinematics and the translation solver are taken from
``oracle/mp_oracle.py``.  Pinned the same way: tests/test_oracle_golden.py checks it against golden G2.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from . import mp_oracle as O


class TorchRNN(nn.Module):
    """models/rnn.py:9-33."""

    def __init__(self, n_input, n_output, n_hidden, bidirectional=True):
        super().__init__()
        self.rnn = nn.LSTM(n_hidden, n_hidden, 2, bidirectional=bidirectional)
        self.linear1 = nn.Linear(n_input, n_hidden)
        self.linear2 = nn.Linear(n_hidden * (2 if bidirectional else 1), n_output)

    def forward(self, x, lengths, h=None):
        data = torch.relu(self.linear1(x))
        data = pack_padded_sequence(data, lengths, batch_first=True, enforce_sorted=False)
        data, h = self.rnn(data, h)
        data, _ = pad_packed_sequence(data, batch_first=True)
        return self.linear2(data), h


class TorchNet:
    def __init__(self, sd, J):
        self.mods = {}
        for name, (n_in, n_out, hid, bi) in {"joints": (60, 72, 256, True), "pose": (132, 96, 256, True),
                                              "foot_contact": (132, 2, 64, True), "velocity": (132, 72, 256, False)}.items():
            m = TorchRNN(n_in, n_out, hid, bi)
            pre = O.PREFIX[name]
            m.load_state_dict({k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith(pre)})
            self.mods[name] = m.eval()
        self.J = np.asarray(J, dtype=np.float32)
        self.floor_y = float((self.J - self.J[:1])[10:12, 1].min())
        self.vel_state = None

    @torch.no_grad()
    def forward(self, imu, lengths):
        """models/net.py:101-119 -> numpy (pose, joints, vel, contact, r6d)."""
        x = torch.from_numpy(np.asarray(imu, dtype=np.float32))
        joints, _ = self.mods["joints"](x, lengths)
        x132 = torch.cat((joints, x), dim=-1)
        r6d, _ = self.mods["pose"](x132, lengths)
        contact, _ = self.mods["foot_contact"](x132, lengths)
        vel, self.vel_state = self.mods["velocity"](x132, lengths, self.vel_state)
        pose = O.reduced_global_to_full(r6d.numpy())
        return pose, joints.numpy(), vel.numpy(), contact.numpy(), r6d.numpy()
