"""ctypes binding of libmobileposer_hip.so (C ABI: include/mobileposer_hip.h).

The product path has no CPU fallback: if the shared library is missing or lacks a symbol this module
raises, and every facade method goes through it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "libmobileposer_hip.so")
LIB_PATH = os.environ.get("MP_LIB_PATH") or _DEFAULT_LIB   # override: kernel-variant A/B runs (no build-id check then)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")


def source_md5():
    """md5 over the sources the library is built from (csrc/*.hip, csrc/*.h, include/*.h, in name order).  The device code is a
    function of these; the build bakes the value into the binary (mp_build_id) and load() compares."""
    import glob
    import hashlib
    m = hashlib.md5()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                    glob.glob(os.path.join(INCLUDE, "*.h"))):
        m.update(os.path.basename(f).encode())
        m.update(open(f, "rb").read())
    return m.hexdigest()


def file_build_id(path=None):
    """The build id baked into a library FILE (without loading it): the text behind the MP_BUILD_ID= marker, or None."""
    try:
        blob = open(path or LIB_PATH, "rb").read()
    except OSError:
        return None
    i = blob.find(b"MP_BUILD_ID=")
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + 12:j].decode(errors="replace")

MP_OK = 0
MP_ERR_INVALID, MP_ERR_HIP, MP_ERR_STATE_SHAPE, MP_ERR_NO_STREAMS, MP_ERR_LENGTHS, MP_ERR_DEVICE = -1, -2, -3, -4, -5, -6
MOD_JOINTS, MOD_POSE, MOD_FOOT_CONTACT, MOD_VELOCITY = 0, 1, 2, 3

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)

# name -> (restype, argtypes): every symbol include/mobileposer_hip.h declares, and the test / debug hooks of
# include/mobileposer_hip_internal.h (mp_set_transport, mp_debug_*)
SIGNATURES = {
    "mp_weight_count": (_sz, []),
    "mp_build_id": (C.c_char_p, []),
    "mp_manifest_entry": (_i, [_i, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_sz)]),
    "mp_create": (_i, [C.POINTER(_vp), _i, _fp, _sz, _ip, _fp]),
    "mp_create_from_device": (_i, [C.POINTER(_vp), _i, _vp, _sz, _ip, _fp]),
    "mp_create_body": (_i, [C.POINTER(_vp), _i, _ip, _fp]),
    "mp_destroy": (None, [_vp]),
    "mp_last_error": (C.c_char_p, [_vp]),
    "mp_get_constants": (_i, [_vp, _fp, _fp]),
    "mp_device_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mp_forward": (_i, [_vp, _vp, _ip, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_forward_offline": (_i, [_vp, _vp, _ip, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_rnn_forward": (_i, [_vp, _i, _vp, _ip, _i, _i, _vp, _vp, _vp, _vp]),
    "mp_reduced_global_to_full": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "mp_inverse_kinematics_r": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "mp_r6d_to_rotation_matrix": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "mp_translate_offline": (_i, [_vp, _vp, _vp, _vp, _ip, _i, _i, _vp, _vp]),
    "mp_fk": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "mp_set_mesh": (_i, [_vp, _fp, _fp, _i]),
    "mp_fk_mesh": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mp_set_shape_space": (_i, [_vp, _fp, _fp]),
    "mp_fk_shape": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mp_zero_pose_body": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mp_set_pose_blendshape": (_i, [_vp, _fp]),
    "mp_eval_metrics": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, C.c_uint, C.c_uint, _i, _vp, _vp]),
    "mp_reset_state": (_i, [_vp, _i]),
    "mp_get_velocity_state": (_i, [_vp, _vp, C.POINTER(_i)]),
    "mp_set_velocity_state": (_i, [_vp, _vp, _i]),
    "mp_stream_create": (_i, [_vp, _i]),
    "mp_stream_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_stream_replay": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mp_stream_reset": (_i, [_vp, C.POINTER(C.c_uint8), _i]),
    "mp_live_form_frames": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint, _i, _vp, _vp]),
    "mp_stream_get_state": (_i, [_vp, _i, _vp, _fp, C.POINTER(C.c_double), _fp, C.POINTER(_i)]),
    "mp_stream_set_state": (_i, [_vp, _i, _vp, _fp, C.POINTER(C.c_double), _fp, C.POINTER(_i)]),
    "mp_timing_enable": (_i, [_vp, _i]),
    "mp_timing_read": (_i, [_vp, _i, C.POINTER(_i), _fp, C.POINTER(C.c_double)]),
    "mp_set_graph_mode": (_i, [_vp, _i]),
    "mp_set_lstm_mode": (_i, [_vp, _i]),
    "mp_set_transport": (_i, [_vp, _i]),
    "mp_device_error": (_i, [_vp, C.POINTER(_i)]),
    "mp_finish": (_i, [_vp]),
    "mp_set_recovery": (_i, [_vp, _i]),
    "mp_recovery_count": (_i, [_vp]),
    "mp_debug_poke_error": (_i, [_vp, _i]),
    "mp_debug_read_prof": (_i, [_vp, C.POINTER(C.c_longlong), _i]),
    "mp_debug_drop_workgroup": (_i, [_vp, _i, _i, _i]),
    "mp_debug_clock_probe": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_debug_clock_probe_loaded": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_debug_plan_stats": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_longlong)]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if LIB_PATH == _DEFAULT_LIB and file_build_id() != source_md5():
        # missing, or built from other sources than the ones beside it: a stale library must be neither timed nor tested.
        # In a SOURCE CHECKOUT (this repository: __graft_entry__.py beside the package) it is rebuilt in place, with a line on
        # stderr that says so (hipcc is part of the image, here and on the GPU box; the build takes a file lock, so ranks that
        # start together build once) -- before the library is mapped into this process.  MP_AUTO_REBUILD=0 refuses instead; an
        # installed package (no __graft_entry__.py of ours beside it) always refuses.
        entry = os.path.join(os.path.dirname(_HERE), "__graft_entry__.py")
        ours = os.path.isfile(entry) and b"mobileposer_amd" in open(entry, "rb").read(4096)
        stale = "mobileposer_amd: %s is missing or stale (build id %r, sources %s)" % (LIB_PATH, file_build_id(), source_md5())
        if os.environ.get("MP_AUTO_REBUILD", "1") == "0" or not ours:
            raise RuntimeError(stale + " -- build it with `python __graft_entry__.py` (hipcc, gfx950).  There is no CPU fallback.")
        try:
            import importlib.util
            import sys
            print(stale + ": rebuilding in place (MP_AUTO_REBUILD=0 refuses instead)", file=sys.stderr)
            spec = importlib.util.spec_from_file_location("__graft_entry__", entry)
            ge = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ge)
            ge.compile_library()
        except Exception as e:                                        # noqa: BLE001
            raise RuntimeError(stale + " and rebuilding it failed: %s -- build it with `python __graft_entry__.py` (hipcc, gfx950).  "
                               "There is no CPU fallback." % e)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "mobileposer_amd: %s not found -- build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; this library must bind to the runtime instance torch uses (the caller's
    # tensors and streams live there), so torch is loaded first when it is installed.  Loaded the other way round, the
    # library resolves the system ROCm's libamdhip64 and its first hipSetDevice fails once torch's copy is in the process.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.mp_build_id().decode()
    if got != source_md5():
        if LIB_PATH == _DEFAULT_LIB:
            raise RuntimeError("mobileposer_amd: %s was built from other sources (build id %s, sources %s)" % (LIB_PATH, got, source_md5()))
        import warnings                      # an MP_LIB_PATH library (A/B runs of kernel variants): allowed, but never silently
        warnings.warn("mobileposer_amd: MP_LIB_PATH library %s has build id %s, the sources beside the package are %s" % (LIB_PATH, got, source_md5()))
    _lib = lib
    return lib


def build_id():
    """mp_build_id() of the loaded library."""
    return load().mp_build_id().decode()


def last_error(handle=None):
    msg = load().mp_last_error(handle)
    return msg.decode() if msg else ""


def check(rc, handle=None):
    if rc != MP_OK:
        raise RuntimeError("libmobileposer_hip: %s (status %d)" % (last_error(handle), rc))
