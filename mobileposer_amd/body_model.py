"""SMPL body model on the GPU: the subset of ``ParametricModel`` (articulate/model.py:20-39,77-92,208-240) that
MobilePoserNet, PoseDataset and the evaluator use -- ``parent``, zero-pose joints / vertices, ``forward_kinematics``
(joints, optional mesh, optional shape).

All arithmetic runs in the HIP library (mp_fk / mp_fk_mesh / mp_fk_shape).  A model that belongs to a MobilePoserNet
uses that net's handle; a stand-alone one (``ParametricModel(paths.smpl_file)`` as data.py:24 and
articulate/evaluator.py:293 build it) owns a body-only handle (mp_create_body).  There is no host implementation.
"""
import ctypes as C
import pickle
import weakref

import numpy as np
import torch

from . import _lib
from .config import SMPL_PARENT


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dense(a):
    return np.asarray(a.toarray() if hasattr(a, "toarray") else a, dtype=np.float32)


class ParametricModel:
    def __init__(self, official_model_file=None, use_pose_blendshape=False, device="cuda:0", data=None):
        """From an SMPL pickle (articulate/model.py:26-37: latin1 pickle, scipy-sparse ``J_regressor``, ``kintree_table``)
        or an already-loaded dict with the same keys."""
        if data is None:
            with open(official_model_file, "rb") as f:
                data = pickle.load(f, encoding="latin1")
        self._J = np.ascontiguousarray(np.asarray(data["J"], dtype=np.float32))
        kt = np.asarray(data["kintree_table"])[0].tolist()
        self.parent = [-1] + [int(p) for p in kt[1:]]          # parent[0] = None in the reference (model.py:37)
        self._v_template = np.asarray(data["v_template"], dtype=np.float32) if "v_template" in data else None
        self._skinning_weights = np.asarray(data["weights"], dtype=np.float32) if "weights" in data else None
        self._shapedirs = np.asarray(data["shapedirs"], dtype=np.float32) if "shapedirs" in data else None
        self._J_regressor = _dense(data["J_regressor"]) if "J_regressor" in data else None
        self._posedirs = np.asarray(data["posedirs"], dtype=np.float32) if "posedirs" in data else None
        self.face = data.get("f")
        # articulate/model.py:38,236-238.  No reference caller enables it (net.py:37, evaluator.py:293, data.py:24)
        self.use_pose_blendshape = bool(use_pose_blendshape)
        if self.use_pose_blendshape and self._posedirs is None:
            raise ValueError("use_pose_blendshape=True needs 'posedirs' in the model file")
        self.device = torch.device(device)
        self._net_ref = None
        self._own = None              # body-only native handle (created on first use when not bound to a net)

    @classmethod
    def synthetic(cls, device="cuda:0"):
        from .synthetic import synthetic_smpl
        return cls(data=synthetic_smpl(), device=device)

    @property
    def J(self):
        return self._J

    def get_zero_pose_joint_and_vertex(self, shape=None):
        """articulate/model.py:77-92.  shape None: root-aligned joints [24,3] and vertices [V,3] of the mean shape (host
        constants, numpy).  shape reshapeable to [S,10]: joints [S,24,3] and vertices [S,V,3] of those bodies, each aligned to
        its own root joint, computed on the GPU (mp_zero_pose_body) and returned as device tensors."""
        if shape is None:
            j = self._J - self._J[:1]
            v = None if self._v_template is None else self._v_template - self._J[:1]
            return j, v
        lib, h, dev = self._handle()
        net = self._net
        state = net._mesh_state if (net is not None and net._h is not None) else self._own_state
        if not state.get("has_shape"):
            raise RuntimeError("the body model has no shape space (shapedirs / J_regressor) loaded")
        sh = torch.as_tensor(shape).to(device=dev, dtype=torch.float32).reshape(-1, 10).contiguous()
        S, V = sh.shape[0], state["n_vertex"]
        j = torch.empty(S, 24, 3, device=dev, dtype=torch.float32)
        v = torch.empty(S, V, 3, device=dev, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mp_zero_pose_body(h, _ptr(sh), S, _ptr(j), _ptr(v), stream), h)
        return j, v

    # ---- native handle ---------------------------------------------------------------------------------------------
    def bind(self, net):
        """Attach the MobilePoserNet whose library handle (holding these constants on the GPU) runs FK.
        Held weakly: the net owns the body model, not the other way round."""
        self._net_ref = weakref.ref(net)

    @property
    def _net(self):
        return self._net_ref() if self._net_ref is not None else None

    def _handle(self):
        """(lib, handle, device) of the native instance that holds this body's constants."""
        net = self._net
        if net is not None and net._h is not None:
            return net._lib, net._h, net.device
        if self._own is None:
            lib = _lib.load()                                  # raises when the HIP library is not built
            if self.device.type != "cuda":
                raise RuntimeError("mobileposer_amd runs on an AMD GPU only (device=%s)" % self.device)
            h = C.c_void_p()
            parent = (C.c_int32 * 24)(*self.parent)
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            _lib.check(lib.mp_create_body(C.byref(h), idx, parent, self._J.reshape(-1).ctypes.data_as(C.POINTER(C.c_float))), None)
            self._own = h
            self._own_state = {}
            self._finalizer = weakref.finalize(self, _destroy, lib, h)
            upload_mesh(lib, h, self, self._own_state)
        return _lib.load(), self._own, self.device

    def close(self):
        h, self._own = self._own, None
        if h is not None:
            self._finalizer.detach()                          # (the handle is released here, not a second time later)
            _lib.load().mp_destroy(h)

    def forward_kinematics(self, pose, shape=None, tran=None, calc_mesh=False):
        """articulate/model.py:208-240 on the GPU.  pose [N,24,3,3] (or reshapeable); shape None | [10] | [1,10] | [N,10];
        tran None | [N,3] -> (R_global [N,24,3,3], joint [N,24,3]) and, with ``calc_mesh``, vertices [N,V,3]."""
        lib, h, dev = self._handle()
        net = self._net
        state = net._mesh_state if (net is not None and net._h is not None) else self._own_state
        return fk_call(lib, h, dev, state, pose, shape, tran, calc_mesh)


    def forward_kinematics_R(self, R_local):
        """articulate/model.py:126-144: local joint rotations [N, *] (reshapeable to [N,24,3,3]) -> global ones [N,24,3,3] -- the
        rotation half of forward_kinematics (mp_fk; the joint positions it also computes are dropped)."""
        R = torch.as_tensor(R_local)
        return self.forward_kinematics(R.reshape(R.shape[0], -1, 3, 3))[0]

    def inverse_kinematics_R(self, R_global):
        """articulate/model.py:146-164: global joint rotations [N, *] (reshapeable to [N,24,3,3]) -> local ones [N,24,3,3],
        R_local[i] = R_global[parent[i]]^T R_global[i] (mp_inverse_kinematics_r)."""
        lib, h, dev = self._handle()
        R = torch.as_tensor(R_global)
        g = R.to(device=dev, dtype=torch.float32).reshape(R.shape[0], 24, 3, 3).contiguous()
        out = torch.empty_like(g)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mp_inverse_kinematics_r(h, _ptr(g), int(g.shape[0]), _ptr(out), stream), h)
        return out


    def eval_metrics(self, pose_p, pose_t, tran_p=None, tran_t=None, fps=60, align_joint=0, joint_mask=None, ignored=()):
        """FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343) on this body: the error table [10,2] in ONE library
        call (mp_eval_metrics) -- what the reference's evaluator computes with the CPU body model it builds from a model file."""
        lib, h, dev = self._handle()
        net = self._net
        state = net._mesh_state if (net is not None and net._h is not None) else self._own_state
        return eval_metrics_call(lib, h, dev, state.get("n_vertex", 0), pose_p, pose_t, tran_p, tran_t, fps, align_joint,
                                 joint_mask, ignored)


def _destroy(lib, h):
    try:
        lib.mp_destroy(h)
    except Exception:
        pass


def upload_mesh(lib, h, bm, state):
    """mp_set_mesh / mp_set_shape_space for the handle ``h`` from the body model's constants; ``state`` records what the
    handle holds (n_vertex, has_shape)."""
    state["n_vertex"], state["has_shape"] = 0, False
    if bm._v_template is None or bm._skinning_weights is None:
        return
    fp = C.POINTER(C.c_float)
    vt = np.ascontiguousarray(bm._v_template, dtype=np.float32)
    sw = np.ascontiguousarray(bm._skinning_weights, dtype=np.float32)
    _lib.check(lib.mp_set_mesh(h, vt.ctypes.data_as(fp), sw.ctypes.data_as(fp), vt.shape[0]), h)
    state["n_vertex"] = int(vt.shape[0])
    if bm._shapedirs is not None and bm._J_regressor is not None:
        sd = np.ascontiguousarray(bm._shapedirs, dtype=np.float32)
        jr = np.ascontiguousarray(bm._J_regressor, dtype=np.float32)
        if sd.shape == (vt.shape[0], 3, 10) and jr.shape == (24, vt.shape[0]):
            _lib.check(lib.mp_set_shape_space(h, sd.ctypes.data_as(fp), jr.ctypes.data_as(fp)), h)
            state["has_shape"] = True
    if getattr(bm, "use_pose_blendshape", False):
        pd = np.ascontiguousarray(bm._posedirs, dtype=np.float32)
        if pd.shape != (vt.shape[0], 3, 207):
            raise ValueError("posedirs must be [V,3,207], got %s" % (pd.shape,))
        _lib.check(lib.mp_set_pose_blendshape(h, pd.ctypes.data_as(fp)), h)


def fk_call(lib, h, dev, state, pose, shape, tran, calc_mesh):
    f32 = torch.float32
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = torch.as_tensor(pose).to(device=dev, dtype=f32).reshape(-1, 24, 3, 3).contiguous()
    N = p.shape[0]
    t = None if tran is None else torch.as_tensor(tran).to(device=dev, dtype=f32).reshape(N, 3).contiguous()
    Rg = torch.empty(N, 24, 3, 3, device=dev, dtype=f32)
    jg = torch.empty(N, 24, 3, device=dev, dtype=f32)
    V = state.get("n_vertex", 0)
    if calc_mesh and not V:
        raise RuntimeError("the body model has no mesh (v_template / weights) loaded")
    vg = torch.empty(N, V, 3, device=dev, dtype=f32) if calc_mesh else None
    if shape is not None:
        if not state.get("has_shape"):
            raise RuntimeError("the body model has no shape space (shapedirs / J_regressor) loaded")
        sh = torch.as_tensor(shape).to(device=dev, dtype=f32).reshape(-1, 10).contiguous()
        if sh.shape[0] not in (1, N):
            raise RuntimeError("shape must expand to [%d, 10], got %s" % (N, tuple(sh.shape)))
        _lib.check(lib.mp_fk_shape(h, _ptr(p), _ptr(sh), int(sh.shape[0]), _ptr(t), N, _ptr(Rg), _ptr(jg), _ptr(vg), stream), h)
    elif calc_mesh:
        _lib.check(lib.mp_fk_mesh(h, _ptr(p), _ptr(t), N, _ptr(Rg), _ptr(jg), _ptr(vg), stream), h)
    else:
        _lib.check(lib.mp_fk(h, _ptr(p), _ptr(t), N, _ptr(Rg), _ptr(jg), stream), h)
    return (Rg, jg, vg) if calc_mesh else (Rg, jg)


def eval_metrics_call(lib, h, dev, n_vertex, pose_p, pose_t, tran_p, tran_t, fps, align_joint, joint_mask, ignored):
    f = lambda t: None if t is None else torch.as_tensor(t).to(device=dev, dtype=torch.float32).contiguous()
    pp, pt = f(pose_p).reshape(-1, 24, 3, 3), f(pose_t).reshape(-1, 24, 3, 3)
    N = int(pp.shape[0])
    if int(pt.shape[0]) != N:
        raise RuntimeError("prediction has %d frames, ground truth %d" % (N, int(pt.shape[0])))
    tp = None if tran_p is None else f(tran_p).reshape(N, 3)
    tt = None if tran_t is None else f(tran_t).reshape(N, 3)

    def bits(js):
        js = [] if js is None else [int(j) for j in js]
        if any(j < 0 or j > 23 for j in js):
            raise ValueError("joint indices must be in 0..23, got %s" % (js,))
        return sum(1 << j for j in set(js))

    table = torch.empty(10, 2, device=dev, dtype=torch.float32)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.mp_eval_metrics(h, _ptr(pp), _ptr(pt), _ptr(tp), _ptr(tt), N, int(fps), int(align_joint), bits(joint_mask),
                                   bits(ignored), int(n_vertex > 0), _ptr(table), stream), h)
    return table


assert SMPL_PARENT[0] == -1
