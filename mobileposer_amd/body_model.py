"""SMPL constants the inference path needs, and the GPU forward kinematics facade.

Mirror of the subset of ``ParametricModel`` (articulate/model.py:20-39,77-92,208-232) used by
MobilePoserNet and the evaluator: ``parent``, zero-pose joints, ``forward_kinematics`` (no mesh).
"""
import pickle
import weakref

import numpy as np

from .config import SMPL_PARENT


class ParametricModel:
    def __init__(self, official_model_file=None, data=None):
        """From an SMPL pickle (articulate/model.py:26-37) or an already-loaded dict with the same keys."""
        if data is None:
            with open(official_model_file, "rb") as f:
                data = pickle.load(f, encoding="latin1")
        self._J = np.asarray(data["J"], dtype=np.float32)
        kt = np.asarray(data["kintree_table"])[0].tolist()
        self.parent = [-1] + [int(p) for p in kt[1:]]          # parent[0] = None in the reference (model.py:37)
        self._v_template = np.asarray(data["v_template"], dtype=np.float32) if "v_template" in data else None
        self._skinning_weights = np.asarray(data["weights"], dtype=np.float32) if "weights" in data else None
        self.face = data.get("f")
        self._net_ref = None

    @classmethod
    def synthetic(cls):
        from .synthetic import synthetic_smpl
        return cls(data=synthetic_smpl())

    @property
    def J(self):
        return self._J

    def get_zero_pose_joint_and_vertex(self):
        """articulate/model.py:77-92 with shape=None: root-aligned joints and vertices."""
        j = self._J - self._J[:1]
        v = None if self._v_template is None else self._v_template - self._J[:1]
        return j, v

    def bind(self, net):
        """Attach the MobilePoserNet whose library handle (holding these constants on the GPU) runs FK.
        Held weakly: the net owns the body model, not the other way round."""
        self._net_ref = weakref.ref(net)

    @property
    def _net(self):
        return self._net_ref() if self._net_ref is not None else None

    def forward_kinematics(self, pose, shape=None, tran=None, calc_mesh=False):
        """articulate/model.py:208-240 on the GPU (mp_fk / mp_fk_mesh).  pose [N,24,3,3] (or reshapeable) cuda tensor."""
        if shape is not None:
            raise NotImplementedError("shape blend shapes are outside the hot path (mean shape only)")
        if self._net is None:
            raise RuntimeError("ParametricModel is not bound to a MobilePoserNet (no GPU handle)")
        return self._net.forward_kinematics(pose, tran, calc_mesh=calc_mesh)


assert SMPL_PARENT[0] == -1
