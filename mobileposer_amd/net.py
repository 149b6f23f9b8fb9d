"""MobilePoserNet facade: the reference's method surface (models/net.py:22-219) over libmobileposer_hip.so.

Drop-in for the inference path: ``MobilePoserNet(...)``, ``load_state_dict``, ``eval``, ``reset``,
``forward``, ``forward_offline``, ``forward_online`` take and return what the reference's methods do
(torch tensors on the GPU), including the quirks listed in SURVEY.md 8(a) (stale velocity state Q1,
batch-size change raises Q2, padded-sequence semantics Q4, raw-logit online weight Q5, ...).
All arithmetic happens in hand-written HIP kernels behind the C ABI; torch only owns device memory.
There is no CPU path: constructing the model without the built library raises.
"""
import atexit
import ctypes as C
import os
import sys
import warnings
import weakref

import numpy as np
import torch

from . import _lib
from .body_model import ParametricModel, eval_metrics_call, fk_call, upload_mesh
from .config import joint_set, model_config, paths
from .manifest import state_dict_manifest
from .model_utils import blob_to_state_dict, state_dict_to_blob


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_LIVE = weakref.WeakSet()       # nets that own a native handle


def _env_graph_mode():
    """MP_GRAPH as csrc/mp_handle.hip reads it: unset, empty or starting with '0' = eager; starting with '2' = single-branch
    graphs; anything else = multi-branch graphs."""
    e = os.environ.get("MP_GRAPH", "")
    if not e or e[0] == "0":
        return 0
    return 2 if e[0] == "2" else 1


@atexit.register
def _close_all():               # runs before interpreter finalisation, while the HIP runtime is still loaded
    for net in list(_LIVE):
        try:
            net.close()
        except Exception:
            pass


class _ModuleView:
    """Stands in for a sub-module of the reference's net -- ``model.joints`` / ``model.pose`` / ``model.foot_contact`` /
    ``model.velocity`` (models/joints.py:48-52, poser.py:60-63, footcontact.py:38-41, velocity.py:40-48): calling it runs that
    block alone (linear1 + ReLU -> 2-layer LSTM, packed -> linear2, models/rnn.py:20-33) through ``mp_rnn_forward`` and returns what
    the reference's ``forward`` returns, the block's output ``[B, T, n_out]``.  With ``input_lengths=None`` the reference feeds
    nn.LSTM time-major data (rnn.py:15,25): dim 0 is time -- reproduced by transposition, as in ``forward``.
    Holds the net weakly: no reference cycle, so a net's native handle is released as soon as the net is."""

    def __init__(self, net, name):
        self._ref = weakref.ref(net)
        self._name = name

    @property
    def _net(self):
        net = self._ref()
        if net is None:
            raise ReferenceError("the MobilePoserNet this %s view belongs to is gone" % self._name)
        return net

    def _run(self, batch, input_lengths, state):
        net = self._net
        if batch.dim() != 3:
            raise RuntimeError("expected a batch of shape [B, T, n_in], got %s" % (tuple(batch.shape),))
        if input_lengths is None:          # dim 0 is time, dim 1 the batch; the returned state has batch T
            B, T = int(batch.shape[0]), int(batch.shape[1])
            xt = batch.to(device=net.device, dtype=torch.float32).transpose(0, 1).contiguous()
            y, st = net.rnn_forward(self._name, xt, [B] * T, state)
            return y.transpose(0, 1).contiguous(), st
        return net.rnn_forward(self._name, batch, input_lengths, state)

    def forward(self, batch, input_lengths=None):
        return self._run(batch, input_lengths, None)[0]

    __call__ = forward

    def eval(self):
        return self


class _VelocityView(_ModuleView):
    """``model.velocity``: the block itself (``forward``: zero initial state, velocity.py:40-43; ``forward_online``: on the carried
    state, which it replaces, velocity.py:45-48) and the carried LSTM state as ``rnn_state`` (velocity.py:30,47) -- ONE state per
    model, the one ``MobilePoserNet.forward`` runs on (net.py:117)."""

    def __init__(self, net):
        super().__init__(net, "velocity")

    def forward_online(self, batch, input_lengths=None):
        vel, state = self._run(batch, input_lengths, self.rnn_state)
        self.rnn_state = state
        return vel

    @property
    def rnn_state(self):
        net = self._net
        b = C.c_int(0)
        net._check(net._lib.mp_get_velocity_state(net._h, None, C.byref(b)))
        if b.value == 0:
            return None
        buf = torch.empty(2, 2, b.value, 256, device=net.device, dtype=torch.float32)
        net._check(net._lib.mp_get_velocity_state(net._h, _ptr(buf), C.byref(b)))
        return buf[0], buf[1]

    @rnn_state.setter
    def rnn_state(self, value):
        net = self._net
        if value is None:
            net._check(net._lib.mp_reset_state(net._h, 1))
            return
        h, c = value
        if h.dim() != 3 or tuple(h.shape) != tuple(c.shape) or int(h.shape[0]) != 2 or int(h.shape[2]) != 256:
            raise RuntimeError("velocity.rnn_state is (h, c), each [2, B, 256] (velocity.py:29-30), got %s / %s" % (tuple(h.shape), tuple(c.shape)))
        buf = torch.stack((h, c)).to(device=net.device, dtype=torch.float32).contiguous()
        net._check(net._lib.mp_set_velocity_state(net._h, _ptr(buf), int(h.shape[1])))


class _PoserView(_ModuleView):
    """``model.pose``: the block (-> 6D rotations of the 16 reduced joints, [B, T, 96]) and ``_reduced_global_to_full``
    (poser.py:52-58 = net.py:93-99)."""

    def __init__(self, net):
        super().__init__(net, "pose")

    def _reduced_global_to_full(self, reduced_pose):
        return self._net._reduced_global_to_full(reduced_pose)


class MobilePoserNet:
    """Inputs: N IMUs.  Outputs: SMPL pose (rotation matrices) and translation.  (models/net.py:22-26)"""

    def __init__(self, poser=None, joints=None, foot_contact=None, velocity=None, finetune=False,
                 smpl_file=None, smpl=None, device="cuda:0"):
        self._lib = _lib.load()                      # raises when the HIP library is not built
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("mobileposer_amd runs on an AMD GPU only (device=%s)" % device)
        self.C = model_config
        self.finetune = finetune
        # body model (net.py:37-38)
        if smpl is not None:
            self.bodymodel = smpl if isinstance(smpl, ParametricModel) else ParametricModel(data=smpl, device=device)
        elif smpl_file is not None:
            self.bodymodel = ParametricModel(smpl_file, device=device)
        elif os.path.exists(str(paths.smpl_file)):             # the reference always reads paths.smpl_file (net.py:37)
            self.bodymodel = ParametricModel(str(paths.smpl_file), device=device)
        else:                                                  # licensed file absent: synthetic body
            self.bodymodel = ParametricModel.synthetic(device=device)
        self.bodymodel.bind(self)
        self.global_to_local_pose = self.bodymodel.inverse_kinematics_R          # net.py:38
        # base joints (net.py:47-49)
        self.j, _ = self.bodymodel.get_zero_pose_joint_and_vertex()
        self.feet_pos = torch.from_numpy(self.j[10:12].copy())
        self.floor_y = float(self.j[10:12, 1].min())
        # constants (net.py:52-56)
        self.gravity_velocity = torch.tensor([0.0, joint_set.gravity_velocity, 0.0], device=self.device)      # net.py:52
        self.prob_threshold = (0.5, 0.9)
        self.num_past_frames = model_config.past_frames
        self.num_future_frames = model_config.future_frames
        self.num_total_frames = self.num_past_frames + self.num_future_frames
        # variables (net.py:59-64): last_root_pos / current_root_y / imu / last_{l,r}foot_pos live in the library's
        # per-stream state on the device and are mirrored by the properties below
        self.rnn_state = None
        self.last_joints = torch.zeros(24, 3, device=self.device)                # net.py:62 (never read by the reference either)
        self.velocity = _VelocityView(self)
        self.joints = _ModuleView(self, "joints")
        self.pose = _PoserView(self)
        self.foot_contact = _ModuleView(self, "foot_contact")
        self._h = None
        self._blob = None
        self._stream_S = 0
        self._graph = _env_graph_mode()      # graph mode of the handle (set_graph_mode / env MP_GRAPH, parsed as the library does)
        self._graph_bufs = {}
        self._tick = 0                  # bumped by everything that changes per-stream state (cache key of stream_state)
        self._state_cache = {}
        self.training = False
        parts = {"pose.": poser, "joints.": joints, "foot_contact.": foot_contact, "velocity.": velocity}
        if any(v is not None for v in parts.values()):
            if not all(v is not None for v in parts.values()):
                raise ValueError("give all four sub-modules or none (combine_weights.py:53)")
            sd = {}
            for prefix, mod in parts.items():
                for k, v in (mod.state_dict() if hasattr(mod, "state_dict") else mod).items():
                    sd[prefix + k] = v
            self.load_state_dict(sd)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_pretrained(cls, model_path, **kw):
        """models/net.py:76-82: the model of a (Lightning) checkpoint, marked for fine-tuning -- the weights through
        ``model_utils.load_model`` (same file formats, same refusal of untrusted pickles; ``kw``: smpl_file / device / smpl / trusted).
        Training itself is out of scope: ``finetune`` is carried, ``hypers`` is not."""
        from .model_utils import load_model
        m = load_model(model_path, **kw)
        m.finetune = True
        return m

    @classmethod
    def from_numpy(cls, state_dict, smpl=None, device="cuda:0"):
        net = cls(smpl=smpl, device=device)
        net.load_state_dict(state_dict)
        return net

    @classmethod
    def from_device_blob(cls, blob, smpl=None, device="cuda:0"):
        """Build from a flat fp32 weight blob already in HBM (after the RCCL broadcast)."""
        net = cls(smpl=smpl, device=device)
        net._create(blob_dev=blob)
        return net

    def _create(self, blob_host=None, blob_dev=None):
        if self._h is not None:
            self._lib.mp_destroy(self._h)
            self._h = None
        parent = (C.c_int32 * 24)(*self.bodymodel.parent)
        J = np.ascontiguousarray(self.bodymodel.J, dtype=np.float32).reshape(-1)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if blob_dev is not None:
            assert blob_dev.is_cuda and blob_dev.dtype == torch.float32 and blob_dev.is_contiguous()
            rc = self._lib.mp_create_from_device(C.byref(h), idx, _ptr(blob_dev), blob_dev.numel(), parent,
                                                 J.ctypes.data_as(C.POINTER(C.c_float)))
            self._blob = blob_dev.detach().cpu().numpy().copy()
        else:
            blob_host = np.ascontiguousarray(blob_host, dtype=np.float32)
            rc = self._lib.mp_create(C.byref(h), idx, blob_host.ctypes.data_as(C.POINTER(C.c_float)), blob_host.size,
                                     parent, J.ctypes.data_as(C.POINTER(C.c_float)))
            self._blob = blob_host
        _lib.check(rc, None)
        self._h = h
        self._recoveries = 0
        self._state_cache = {}
        # (the library reads MP_LSTM_MODE at creation, csrc/mp_api.hip create_common: mirror it -- ADVICE r5)
        self._lstm_mode = {"x3": 3, "step": 0}.get(os.environ.get("MP_LSTM_MODE", ""), 1)
        _LIVE.add(self)
        fy = C.c_float()
        fp = (C.c_float * 6)()
        self._check(self._lib.mp_get_constants(self._h, C.byref(fy), fp))
        assert abs(fy.value - self.floor_y) < 1e-6
        self._stream_S = 0
        # Python owns the graph mode: a fresh handle re-reads MP_GRAPH in C, which may differ from what set_graph_mode() last
        # chose -- and a library that captures while the facade hands it fresh tensors every call captures on every call
        self._check(self._lib.mp_set_graph_mode(self._h, int(self._graph)))
        self._graph_bufs = {}
        self._mesh_state = {}
        upload_mesh(self._lib, self._h, self.bodymodel, self._mesh_state)
        self.n_vertex = self._mesh_state["n_vertex"]

    def close(self):
        """Release the native handle (weights, workspaces, streams, graphs).  Idempotent.  Raises if a device error of
        an earlier call was never reported (possible only with recovery off: the affected outputs were NaN)."""
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            rc = self._lib.mp_finish(h)
            msg = _lib.last_error(h) if rc else ""
            self._lib.mp_destroy(h)
            if rc and not sys.is_finalizing():
                raise RuntimeError("libmobileposer_hip: %s (status %d)" % (msg, rc))

    def _check(self, rc):
        """Raise on a failed library call.  A reported device error resets device-side state (mp_recovery.hip
        invalidate_carried_state: streams reset, velocity state dropped), so the read-back cache of the state attributes
        must not survive it."""
        if rc != _lib.MP_OK:
            self._tick += 1
            self._state_cache = {}
        _lib.check(rc, self._h)

    def _after_call(self):
        """A call that was repaired by the library's recovery path (include/mobileposer_hip.h, mp_set_recovery) is
        reported as a Python warning: its results are valid, but the GPU is evidently shared."""
        n = self._lib.mp_recovery_count(self._h)
        if n != self._recoveries:
            self._recoveries = n
            warnings.warn(_lib.last_error(self._h), RuntimeWarning, stacklevel=3)

    def set_recovery(self, on):
        """True (default): every network call waits for itself and repairs a starved fused-LSTM launch by re-running with
        per-step kernels; False: asynchronous calls, errors surface at the next call / ``finish()`` / ``close()``."""
        self._check(self._lib.mp_set_recovery(self._h, int(bool(on))))

    def finish(self):
        """Wait for everything enqueued; raises if a persistent kernel gave up a wait (recovery off)."""
        self._check(self._lib.mp_finish(self._h))

    @property
    def recovery_count(self):
        return int(self._lib.mp_recovery_count(self._h)) if self._h is not None else 0

    def device_info(self):
        """dict(device, n_cu, xcd_round_robin, placement_tables, build_id): what the handle found on its device (mp_device_info)."""
        d, n, x = C.c_int(-1), C.c_int(0), C.c_int(0)
        self._check(self._lib.mp_device_info(self._h, C.byref(d), C.byref(n), C.byref(x)))
        return {"device": d.value, "n_cu": n.value, "xcd_round_robin": bool(x.value & 1), "placement_tables": bool(x.value & 2),
                "build_id": _lib.build_id()}

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            self.close()
        except RuntimeError as e:
            if exc_type is None:
                raise
            # an exception from the body is already propagating: do not mask it with the close()-time device error
            warnings.warn("MobilePoserNet.close(): %s" % e, RuntimeWarning, stacklevel=2)

    def __del__(self):
        # no reference cycles (the velocity view and the body model hold this object weakly), so this runs when the
        # last reference goes -- never from the cyclic collector at an arbitrary allocation.  At interpreter shutdown
        # the HIP runtime may already be gone: handles still alive then were released by the atexit hook above.
        try:
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ torch.nn.Module look-alikes
    def load_state_dict(self, state_dict, strict=True):
        self._create(blob_host=state_dict_to_blob(state_dict))
        return self

    def state_dict(self):
        self._require_weights()
        return {k: torch.from_numpy(v) for k, v in blob_to_state_dict(self._blob).items()}

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device and torch.device(device).type == "cuda" \
                and torch.device(device).index not in (None, self.device.index):
            raise RuntimeError("a MobilePoserNet handle is bound to %s; build another instance for %s" % (self.device, device))
        return self

    def _require_weights(self):
        if self._h is None:
            raise RuntimeError("MobilePoserNet has no weights: call load_state_dict() / load_model() first "
                               "(the reference's random initialisation is not reproduced)")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ reference methods
    def reset(self):
        """models/net.py:84-88.  Like the reference this does NOT clear ``velocity.rnn_state``;
        pass ``clear_velocity=True`` to ``reset_all`` or set ``model.velocity.rnn_state = None`` for that."""
        self.rnn_state = None
        if self._h is not None and self._stream_S:        # imu = None, current_root_y = 0, last_root_pos = 0
            self._check(self._lib.mp_stream_reset(self._h, None, 0))
            self._tick += 1

    def reset_all(self, clear_velocity=True):
        self.reset()
        if self._h is not None and clear_velocity:
            self._check(self._lib.mp_reset_state(self._h, 1))
            if self._stream_S:
                self._check(self._lib.mp_stream_reset(self._h, None, 1))

    def _prob_to_weight(self, p):
        lo, hi = self.prob_threshold
        return (p.clamp(lo, hi) - lo) / (hi - lo)

    def _lengths(self, input_lengths, B, T):
        if input_lengths is None:      # (forward / forward_offline handle None themselves -- the time-major quirk)
            raise ValueError("input_lengths is required here: with None the reference feeds nn.LSTM time-major data "
                             "(rnn.py:15,25; SURVEY Q3); only forward / forward_offline reproduce that")
        lens = [int(x) for x in input_lengths]
        if len(lens) != B:
            raise RuntimeError("len(input_lengths) = %d but batch = %d" % (len(lens), B))
        return (C.c_int32 * B)(*lens)

    def _input(self, x, graph_key=None):
        """The caller's tensor itself when it already is contiguous fp32 on this device (no copy).  Graph mode
        (``graph_key`` given): a copy in a buffer that stays, so that the captured graph's input address does."""
        if self._graph and graph_key is not None:
            held = self._graph_bufs.setdefault(graph_key, {})
            if "x" not in held:
                held["x"] = torch.empty(tuple(x.shape), device=self.device, dtype=torch.float32)
            held["x"].copy_(x)
            return held["x"]
        if x.device == self.device and x.dtype == torch.float32 and x.is_contiguous():
            return x
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _outputs(self, B, T, names):
        """Output tensors the library writes directly.  Eager launches: fresh tensors per call.  Graph mode: one set per
        (B, T), so that a captured graph (keyed by its buffer addresses) is replayed call after call; the caller gets
        clones (``_result``)."""
        shapes = {"pose": (B * T, 24, 3, 3), "joints": (B, T, 72), "vel": (B, T, 72), "contact": (B, T, 2),
                  "r6d": (B, T, 96), "tran": (B, T, 3)}
        if not self._graph:
            return {n: torch.empty(shapes[n], device=self.device, dtype=torch.float32) for n in names}
        key = (B, T)
        held = self._graph_bufs.setdefault(key, {})
        for n in names:
            if n not in held:
                held[n] = torch.empty(shapes[n], device=self.device, dtype=torch.float32)
        if len(self._graph_bufs) > 8:                    # shapes keep changing: forget the oldest (the library does the same)
            self._graph_bufs.pop(next(iter(self._graph_bufs)))
        return {n: held[n] for n in names}

    def _result(self, t):
        return t.clone() if self._graph else t

    def forward_into(self, imu, lengths_c, pose, joints, vel, contact, r6d=None):
        """mp_forward on caller-owned contiguous fp32 cuda buffers (no allocation, no copies)."""
        B, T = imu.shape[0], imu.shape[1]
        rc = self._lib.mp_forward(self._h, _ptr(imu), lengths_c, B, T, _ptr(pose), _ptr(joints), _ptr(vel),
                                  _ptr(contact), _ptr(r6d), self._stream())
        self._check(rc)
        self._after_call()

    def forward(self, batch, input_lengths=None, return_r6d=False):
        """models/net.py:101-119 -> (pred_pose [B*T,24,3,3], pred_joints [B,T,72], pred_vel [B,T,72]
        (batch dim squeezed when B == 1, net.py:117), foot_contact [B,T,2]); with ``return_r6d`` the Poser output
        before net.py:110 ([B,T,96]) is appended.  Outputs are fresh tensors the library writes directly."""
        self._require_weights()
        if batch.dim() != 3 or batch.shape[-1] != model_config.n_imu:
            raise RuntimeError("expected batch of shape [B, T, 60], got %s" % (tuple(batch.shape),))
        B, T = int(batch.shape[0]), int(batch.shape[1])
        if input_lengths is None:
            return self._forward_time_major(batch, B, T, return_r6d)
        lens = self._lengths(input_lengths, B, T)
        x = self._input(batch, (B, T))
        o = self._outputs(B, T, ("pose", "joints", "vel", "contact") + (("r6d",) if return_r6d else ()))
        self.forward_into(x, lens, o["pose"], o["joints"], o["vel"], o["contact"], o.get("r6d"))
        o = {k: self._result(v) for k, v in o.items()}
        out = (o["pose"], o["joints"], o["vel"].squeeze(0), o["contact"])
        return out + (o["r6d"],) if return_r6d else out

    def _forward_time_major(self, batch, B, T, return_r6d):
        """forward(batch, None) as the reference computes it (no reference caller does this): nn.LSTM is built without
        batch_first (rnn.py:15) and only the packed path is batch-first (rnn.py:25), so dim 0 of [B,T,60] is TIME and dim 1 the
        batch -- T sequences of B steps.  The same call on the transposed input, the outputs transposed back; the carried
        velocity state then has batch T, as the reference's."""
        xt = batch.to(device=self.device, dtype=torch.float32).transpose(0, 1).contiguous()
        out = self.forward(xt, [B] * T, return_r6d=True)
        pose_t, joints_t, vel_t, contact_t, r6d_t = out
        back = lambda t: t.reshape(T, B, -1).transpose(0, 1).contiguous()
        pose = pose_t.reshape(T, B, 24, 3, 3).transpose(0, 1).reshape(B * T, 24, 3, 3).contiguous()
        res = (pose, back(joints_t), back(vel_t).squeeze(0), back(contact_t))
        return res + (back(r6d_t),) if return_r6d else res

    __call__ = forward

    @torch.no_grad()
    def forward_offline(self, imu, input_lengths=None):
        """models/net.py:121-171 (PHYSICS off): one sequence [1,T,60] ->
        (pose [T,24,3,3], pred_joints [1,T,72], tran [T,3], contact [T,2]).
        With a batch B > 1 (a generalisation the reference does not have) the leading dims are kept:
        pose [B*T,24,3,3], pred_joints [B,T,72], tran [B,T,3], contact [B,T,2]."""
        self._require_weights()
        if imu.dim() != 3 or imu.shape[-1] != model_config.n_imu:
            raise RuntimeError("expected imu of shape [B, T, 60], got %s" % (tuple(imu.shape),))
        B, T = int(imu.shape[0]), int(imu.shape[1])
        if input_lengths is None:
            # net.py:121-155 on forward(imu, None): one "sequence" [1,T,60] is T one-step sequences, then the solver
            # over the T frames as ever
            if B != 1:
                raise RuntimeError("forward_offline(imu, None) takes one sequence [1, T, 60] (net.py:126 squeezes the batch)")
            pose, joints, vel, contact = self.forward(imu, None)
            tran = torch.empty(1, T, 3, device=self.device, dtype=torch.float32)
            self.translate_offline_into(joints.contiguous(), vel.reshape(1, T, 72).contiguous(), contact.contiguous(),
                                        (C.c_int32 * 1)(T), tran)
            return pose, joints, tran[0], contact[0]
        lens = self._lengths(input_lengths, B, T)
        x = self._input(imu, (B, T))
        o = self._outputs(B, T, ("pose", "joints", "vel", "contact", "tran"))
        rc = self._lib.mp_forward_offline(self._h, _ptr(x), lens, B, T, _ptr(o["pose"]), _ptr(o["joints"]),
                                          _ptr(o["vel"]), _ptr(o["contact"]), _ptr(o["tran"]), None, None, self._stream())
        self._check(rc)
        self._after_call()
        o = {k: self._result(v) for k, v in o.items()}
        if B == 1:
            return o["pose"], o["joints"], o["tran"][0], o["contact"][0]
        return o["pose"], o["joints"], o["tran"], o["contact"]

    def forward_offline_into(self, imu, lengths_c, pose, joints, vel, contact, tran, rglobal=None, joint_global=None):
        """mp_forward_offline on caller-owned contiguous fp32 cuda buffers: forward + translation solver and, when ``rglobal``
        [B*T,24,3,3] / ``joint_global`` [B*T,24,3] are given, SMPL forward kinematics of the predicted pose in the same call
        (articulate/model.py:208-232) -- the call bench.py times."""
        B, T = imu.shape[0], imu.shape[1]
        rc = self._lib.mp_forward_offline(self._h, _ptr(imu), lengths_c, B, T, _ptr(pose), _ptr(joints), _ptr(vel),
                                          _ptr(contact), _ptr(tran), _ptr(rglobal), _ptr(joint_global), self._stream())
        self._check(rc)
        self._after_call()

    def translate_offline_into(self, joints, vel, contact, lengths_c, tran):
        B, T = joints.shape[0], joints.shape[1]
        rc = self._lib.mp_translate_offline(self._h, _ptr(joints), _ptr(vel), _ptr(contact), lengths_c, B, T,
                                            _ptr(tran), self._stream())
        self._check(rc)

    # ---- streaming ---------------------------------------------------------------------------
    def stream_create(self, S):
        self._require_weights()
        self._check(self._lib.mp_stream_create(self._h, int(S)))
        self._stream_S = int(S)

    def stream_step_into(self, frames, pose, joints, root, contact):
        rc = self._lib.mp_stream_step(self._h, _ptr(frames), _ptr(pose), _ptr(joints), _ptr(root), _ptr(contact),
                                      self._stream())
        self._check(rc)
        self._tick += 1
        self._after_call()

    def stream_step(self, frames):
        """One tick for all S streams: frames [S,60] -> (pose [S,24,9], joints [S,45,72], root_pos [S,3], contact [S,2])."""
        S, dev, f32 = self._stream_S, self.device, torch.float32
        x = self._input(frames.reshape(S, 60))
        if self._graph:                                   # replayed graph: fixed buffers in, clones out
            io = self._graph_bufs.setdefault(("stream", S), {})
            if not io:
                io.update(x=torch.empty(S, 60, device=dev, dtype=f32), pose=torch.empty(S, 24, 9, device=dev, dtype=f32),
                          joints=torch.empty(S, 45, 72, device=dev, dtype=f32), root=torch.empty(S, 3, device=dev, dtype=f32),
                          contact=torch.empty(S, 2, device=dev, dtype=f32))
            io["x"].copy_(x)
            self.stream_step_into(io["x"], io["pose"], io["joints"], io["root"], io["contact"])
            return io["pose"].clone(), io["joints"].clone(), io["root"].clone(), io["contact"].clone()
        pose = torch.empty(S, 24, 9, device=dev, dtype=f32)
        joints = torch.empty(S, 45, 72, device=dev, dtype=f32)
        root = torch.empty(S, 3, device=dev, dtype=f32)
        contact = torch.empty(S, 2, device=dev, dtype=f32)
        self.stream_step_into(x, pose, joints, root, contact)
        return pose, joints, root, contact

    def live_form_frames(self, quat, acc, smpl2imu, device2bone, acc_offsets, keep_mask):
        """mp_live_form_frames: raw sensor samples of S streams -> frames [S,60] (live_demo.py:213-236 per stream)."""
        self._require_weights()
        f = lambda t: self._input(torch.as_tensor(t, dtype=torch.float32))
        q, a, M, D, O = f(quat), f(acc), f(smpl2imu), f(device2bone), f(acc_offsets)
        S = int(q.shape[0])
        assert tuple(q.shape) == (S, 5, 4) and tuple(a.shape) == (S, 5, 3) and tuple(M.shape) == (S, 3, 3)
        assert tuple(D.shape) == (S, 5, 3, 3) and O.numel() == S * 15
        out = torch.empty(S, 60, device=self.device, dtype=torch.float32)
        self._check(self._lib.mp_live_form_frames(self._h, _ptr(q), _ptr(a), _ptr(M), _ptr(D), _ptr(O), int(keep_mask), S,
                                                 _ptr(out), self._stream()))
        return out

    def stream_reset(self, mask=None, clear_velocity=False):
        """reset() (net.py:84-88) for the streams whose ``mask`` entry is true (all when None)."""
        m = None
        if mask is not None:
            m = (C.c_uint8 * self._stream_S)(*[1 if bool(v) else 0 for v in mask])
        self._check(self._lib.mp_stream_reset(self._h, m, int(bool(clear_velocity))))
        self._tick += 1

    @torch.no_grad()
    def forward_online(self, data, input_lengths=None):
        """models/net.py:173-219 (PHYSICS off): data [60] ->
        (pose [24,9], pred_joints [45,72], last_root_pos [3], contact [2])."""
        self._require_weights()
        if self._stream_S == 0:
            self.stream_create(1)
        if self._stream_S != 1:
            raise RuntimeError("forward_online drives a single stream; use stream_step for %d streams" % self._stream_S)
        pose, joints, root, contact = self.stream_step(data.reshape(1, 60))
        return pose[0], joints[0], root[0], contact[0]

    @torch.no_grad()
    def forward_online_replay(self, frames):
        """``[self.forward_online(f) for f in frames]`` (evaluate.py:62-64) as ONE library call (mp_stream_replay): frames
        [N,60] -> (pose [N,24,9], pred_joints [N,45,72], last_root_pos [N,3], contact [N,2]), row k = what the k-th call
        returns.  State (window, velocity LSTM state, foot / root state) is left as the last call leaves it."""
        self._require_weights()
        if self._stream_S == 0:
            self.stream_create(1)
        if self._stream_S != 1:
            raise RuntimeError("forward_online_replay drives a single stream; %d were created" % self._stream_S)
        if getattr(self, "_lstm_mode", 1) == 3:           # (the replay runs on exact-fp32 operands; mode 3: the calls one by one)
            outs = [self.forward_online(f) for f in frames.reshape(-1, 60)]
            return tuple(torch.stack([o[i] for o in outs]) for i in range(4))
        x = self._input(frames.reshape(-1, 60))
        N, dev, f32 = int(x.shape[0]), self.device, torch.float32
        pose = torch.empty(N, 24, 9, device=dev, dtype=f32)
        joints = torch.empty(N, 45, 72, device=dev, dtype=f32)
        root = torch.empty(N, 3, device=dev, dtype=f32)
        contact = torch.empty(N, 2, device=dev, dtype=f32)
        # ONE call for the whole sequence (round 6): the library keeps its workspaces by capacity class (csrc/mp_plans.hip get_plan),
        # so a sequence of a length never seen before runs on the plan the longest one so far left behind (round 5 cut the
        # sequence into power-of-two chunks here because every (batch, length) shape had workspaces of its own)
        self._check(self._lib.mp_stream_replay(self._h, _ptr(x), N, _ptr(pose), _ptr(joints), _ptr(root), _ptr(contact), self._stream()))
        self._tick += 1
        self._after_call()
        return pose, joints, root, contact

    # ---- the reference's state attributes (net.py:59-64,205-208), read back from the device ---------------
    def stream_state(self, s=0):
        """State of stream ``s``: dict(imu [45,60] or None before the first frame, current_root_y (float),
        last_root_pos [3], last_lfoot_pos [3], last_rfoot_pos [3]).  Read back once per tick: repeated attribute reads
        between two ticks share one round trip to the device."""
        if self._h is None or not self._stream_S:
            feet = self.feet_pos.to(self.device)
            return {"imu": None, "current_root_y": 0, "last_root_pos": torch.zeros(3, device=self.device),
                    "last_lfoot_pos": feet[0], "last_rfoot_pos": feet[1]}
        hit = self._state_cache.get(s)
        if hit is not None and hit[0] == self._tick:
            # copies: in-place edits of a returned tensor must not look as if they had reached the device -- only
            # ASSIGNMENT to the attribute writes through (mp_stream_set_state), as documented on the properties below
            return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hit[1].items()}
        win = torch.empty(45, 60, device=self.device, dtype=torch.float32)
        feet = (C.c_float * 6)()
        root = (C.c_float * 3)()
        y, fresh = C.c_double(0), C.c_int(0)
        self._check(self._lib.mp_stream_get_state(self._h, int(s), _ptr(win), feet, C.byref(y), root, C.byref(fresh)))
        t = lambda a: torch.tensor(list(a), device=self.device, dtype=torch.float32)
        st = {"imu": None if fresh.value else win, "current_root_y": y.value if not fresh.value else 0,
              "last_root_pos": t(root), "last_lfoot_pos": t(feet[0:3]), "last_rfoot_pos": t(feet[3:6])}
        self._state_cache[s] = (self._tick, st)
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}

    def _set_stream_state(self, name, value, s=0):
        """Assignment to one of the reference's state attributes (net.py:59-64) -> mp_stream_set_state."""
        self._require_weights()
        if self._stream_S == 0:
            self.stream_create(1)
        win = feet = y = root = fresh = None
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in torch.as_tensor(v).reshape(3).tolist()])
        if name == "imu":
            if value is None:
                fresh = C.c_int(1)
            else:
                win = torch.as_tensor(value).to(device=self.device, dtype=torch.float32).reshape(45, 60).contiguous()
                fresh = C.c_int(0)
        elif name == "current_root_y":
            y = C.c_double(float(value))
        elif name == "last_root_pos":
            root = f3(value)
        else:                                   # one foot: read the pair, replace one half
            cur = self.stream_state(s)
            l = value if name == "last_lfoot_pos" else cur["last_lfoot_pos"]
            r = value if name == "last_rfoot_pos" else cur["last_rfoot_pos"]
            feet = (C.c_float * 6)(*(list(f3(l)) + list(f3(r))))
        self._check(self._lib.mp_stream_set_state(self._h, int(s), _ptr(win), feet, C.byref(y) if y is not None else None,
                                                 root, C.byref(fresh) if fresh is not None else None))
        self._tick += 1                         # invalidates the read-back cache

    # Reads return COPIES of the device state (one round trip per tick, shared by the five attributes); assignment
    # (``model.last_root_pos = t``, ``model.imu = None``) writes through to the device.  In-place edits of a value read
    # earlier (``model.last_root_pos[1] = 0``) change only that copy -- assign the edited tensor back.
    def _state_property(name):
        return property(lambda self: self.stream_state()[name], lambda self, v: self._set_stream_state(name, v))

    imu = _state_property("imu")
    current_root_y = _state_property("current_root_y")
    last_root_pos = _state_property("last_root_pos")
    last_lfoot_pos = _state_property("last_lfoot_pos")
    last_rfoot_pos = _state_property("last_rfoot_pos")
    del _state_property

    # ---- kinematics ----------------------------------------------------------------------------
    def _reduced_global_to_full(self, reduced_pose):
        """models/net.py:93-99: 6D global rotations of the 16 reduced joints [..., 96] -> local [N,24,3,3]."""
        self._require_weights()
        r = reduced_pose.to(device=self.device, dtype=torch.float32).reshape(-1, 96).contiguous()
        out = torch.empty(r.shape[0], 24, 3, 3, device=self.device, dtype=torch.float32)
        self._check(self._lib.mp_reduced_global_to_full(self._h, _ptr(r), r.shape[0], _ptr(out), self._stream()))
        return out

    def forward_kinematics(self, pose, tran=None, calc_mesh=False, shape=None):
        """articulate/model.py:208-240: local pose [N,24,3,3] -> (R_global [N,24,3,3], joint [N,24,3])
        and, with ``calc_mesh``, the skinned vertices [N,V,3] (no pose blendshape); ``shape`` [10] | [1,10] | [N,10]
        selects the body (None = mean shape)."""
        self._require_weights()
        return fk_call(self._lib, self._h, self.device, self._mesh_state, pose, shape, tran, calc_mesh)

    def r6d_to_rotation_matrix(self, r6d):
        """art.math.r6d_to_rotation_matrix (angular.py:167-182): [..., 6] -> [N, 3, 3] (mp_r6d_to_rotation_matrix)."""
        self._require_weights()
        r = torch.as_tensor(r6d).to(device=self.device, dtype=torch.float32).reshape(-1, 6).contiguous()
        out = torch.empty(r.shape[0], 3, 3, device=self.device, dtype=torch.float32)
        self._check(self._lib.mp_r6d_to_rotation_matrix(self._h, _ptr(r), r.shape[0], _ptr(out), self._stream()))
        return out

    def eval_metrics(self, pose_p, pose_t, tran_p=None, tran_t=None, fps=60, align_joint=0, joint_mask=None, ignored=()):
        """mp_eval_metrics: FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343) -> error table [10,2] on the
        device, in ONE library call (identity on ``ignored`` joints, FK + skinning of both poses, all ten metrics)."""
        self._require_weights()
        table = eval_metrics_call(self._lib, self._h, self.device, self.n_vertex, pose_p, pose_t, tran_p, tran_t, fps,
                                  align_joint, joint_mask, ignored)
        self._after_call()
        return table

    def rnn_forward(self, module, x, input_lengths, state=None):
        """RNN.forward of one sub-module (models/rnn.py:20-33): -> (y [B,T,n_out], (h_n, c_n))."""
        self._require_weights()
        mod = {"joints": 0, "pose": 1, "foot_contact": 2, "velocity": 3}[module]
        n_out, H, dirs = {0: (72, 256, 2), 1: (96, 256, 2), 2: (2, 64, 2), 3: (72, 256, 1)}[mod]
        n_in = model_config.n_imu if mod == 0 else model_config.n_imu + 72          # joints: imu; the others: cat(pred_joints, imu)
        if x.dim() != 3 or int(x.shape[-1]) != n_in:
            raise RuntimeError("%s takes [B, T, %d], got %s" % (module, n_in, tuple(x.shape)))
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, T = int(x.shape[0]), int(x.shape[1])
        lens = self._lengths(input_lengths, B, T)
        y = torch.empty(B, T, n_out, device=self.device, dtype=torch.float32)
        st_out = torch.empty(2, 2 * dirs, B, H, device=self.device, dtype=torch.float32)
        st_in = None
        if state is not None:
            want = (2 * dirs, B, H)
            for name, t in (("h0", state[0]), ("c0", state[1])):
                if tuple(t.shape) != want:       # (nn.LSTM raises "Expected hidden size ..." here: SURVEY Q2, a batch-size change)
                    raise RuntimeError("Expected %s of shape %s for a batch of %d sequences, got %s" % (name, want, B, tuple(t.shape)))
            st_in = torch.stack((state[0], state[1])).to(device=self.device, dtype=torch.float32).contiguous()
        rc = self._lib.mp_rnn_forward(self._h, mod, _ptr(x), lens, B, T, _ptr(y), _ptr(st_in), _ptr(st_out), self._stream())
        self._check(rc)
        self._after_call()
        return y, (st_out[0], st_out[1])

    # ---- measurement hooks -----------------------------------------------------------------------
    def timing_enable(self, on):
        self._check(self._lib.mp_timing_enable(self._h, int(bool(on))))

    def timing_read(self, cls):
        """(launches, event-measured ms, algorithmic GFLOP) of a kernel class in the last timed call."""
        n, ms, gf = C.c_int(0), C.c_float(0), C.c_double(0)
        self._check(self._lib.mp_timing_read(self._h, cls, C.byref(n), C.byref(ms), C.byref(gf)))
        return n.value, ms.value, gf.value

    def set_lstm_mode(self, mode):
        """1 (default): fused persistent layers on exact-fp32 MFMA operands; 3: the same on split-fp16 operands (opt-in
        fast mode); 2: mode 1 + two-layer wavefront velocity kernel; 0: per-step kernels.  (include/mobileposer_hip.h)"""
        self._check(self._lib.mp_set_lstm_mode(self._h, int(mode)))
        self._lstm_mode = int(mode)

    def set_transport(self, force_remote):
        """Test hook: force the any-placement (sc1) hidden-state transport of the persistent kernels."""
        self._check(self._lib.mp_set_transport(self._h, int(bool(force_remote))))

    def device_error(self):
        """0 = ok; otherwise 1+step at which a persistent-kernel wait timed out (synchronises)."""
        code = C.c_int(0)
        self._check(self._lib.mp_device_error(self._h, C.byref(code)))
        if code.value:                        # reported: the library has reset the carried state (see _check)
            self._tick += 1
            self._state_cache = {}
        return code.value

    def set_graph_mode(self, on):
        """0 / False: eager launches (default); 1 / True: replay captured hipGraphs (multi-branch); 2: single-branch graphs
        (every launch on one stream: nothing for the runtime's graph executor to mis-assign; include/mobileposer_hip.h)."""
        mode = int(on)
        self._check(self._lib.mp_set_graph_mode(self._h, mode))
        self._graph = mode
        self._graph_bufs = {}


assert joint_set.n_reduced == 16 and len(state_dict_manifest()) == 72
