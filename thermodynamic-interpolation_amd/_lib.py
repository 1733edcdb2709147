"""ctypes binding of libti_hip.so (the C ABI of include/ti_hip.h).

Fails loudly: there is no CPU fallback.  If the shared library is missing it is NOT silently replaced by anything --
call ``build.build()`` (needs hipcc) or ``__graft_entry__.build()`` first.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# TI_LIB_PATH: explicit path of an alternative build of the library (tools/variant_bench.py builds experiment variants next to
# the product library instead of over it).  Unset in normal use.
SO_PATH = os.environ.get("TI_LIB_PATH") or os.path.join(HERE, "libti_hip.so")

TI_OK, TI_E_ARG, TI_E_HIP, TI_E_NAN, TI_E_ALLOC, TI_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1
SCHEMES = {"euler": 0, "heun": 1, "em": 2, "dopri5": 3, "midpoint": 4, "rk4": 5}
PRECISIONS = {"f32": 0, "f16x2": 1, "f16": 2}
KERNELS = {"painn_edge": 0, "painn_update": 1, "painn_embed": 2, "painn_readout": 3, "adw": 4, "integrate": 5,
           "painn_jvp_edge": 6, "painn_jvp_update": 7, "painn_jvp_readout": 8, "painn_jvp_filter": 9}

# every symbol include/ti_hip.h declares (tests/test_abi.py checks the library exports exactly these)
ABI_SYMBOLS = [
    "ti_rollout_rows", "ti_version", "ti_device_count", "ti_last_error",
    "ti_adw_create", "ti_adw_drift", "ti_adw_drift_div", "ti_adw_rollout", "ti_adw_rollout_dlogp",
    "ti_painn_create", "ti_painn_drift", "ti_painn_rollout", "ti_painn_drift_jvp", "ti_painn_drift_div", "ti_painn_rollout_dlogp",
    "ti_destroy", "ti_set_stream", "ti_wait_stream", "ti_painn_set_template", "ti_painn_template_for", "ti_reserve", "ti_profile_enable", "ti_profile_read",
    "ti_painn_debug_tap", "ti_painn_debug_read", "ti_painn_debug_poison", "ti_selftest",
]


class PainnDesc(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_features", C.c_int32), ("n_layers", C.c_int32), ("n_types", C.c_int32),
                ("n_atoms", C.c_int32), ("n_edges", C.c_int32), ("temp_length", C.c_float), ("time_length", C.c_float),
                ("length_scale", C.c_float), ("temp_mean", C.c_float), ("temp_range", C.c_float), ("precision", C.c_int32)]


class AdwDesc(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("precision", C.c_int32)]


class RolloutDesc(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("n_step", C.c_int32), ("save_every", C.c_int32), ("mem", C.c_int32),
                ("eps", C.c_float), ("com_free_noise", C.c_int32), ("seed", C.c_uint64), ("traj_offset", C.c_int64),
                ("t_grid", C.POINTER(C.c_float)), ("rtol", C.c_float), ("atol", C.c_float), ("step_offset", C.c_int64)]


class TiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libti_hip error {code}: {msg}")
        self.code = code


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAMEs as /opt/rocm's).  Two HIP
    runtimes in one process cannot both open the GPU ("No HIP GPUs are available" for whichever comes second), so when
    torch is installed libti_hip.so is bound to torch's copy: loading it first (RTLD_GLOBAL) makes the dynamic loader
    resolve libti_hip.so's NEEDED libamdhip64.so.7 to the already-loaded object.  torch itself is not imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libti_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc, gfx950).  There is no CPU fallback for the sampling path.")
    _preload_torch_hip_runtime()
    L = C.CDLL(SO_PATH)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.ti_rollout_rows.restype = C.c_int64
    L.ti_rollout_rows.argtypes = [C.c_int32, C.c_int32]
    L.ti_last_error.restype = C.c_char_p
    L.ti_painn_create.restype = vp
    L.ti_painn_create.argtypes = [C.POINTER(PainnDesc), fp, C.c_size_t, ip, ip, ip, ip, C.c_int]
    L.ti_painn_drift.argtypes = [vp, vp, C.c_float, vp, C.c_int64, vp, C.c_int]
    L.ti_painn_rollout.argtypes = [vp, C.POINTER(RolloutDesc), vp, vp, C.c_int64, vp, C.POINTER(C.c_int64)]
    L.ti_painn_drift_jvp.argtypes = [vp, vp, vp, C.c_float, vp, C.c_int64, vp, vp, C.c_int]
    L.ti_painn_drift_div.argtypes = [vp, vp, C.c_float, vp, C.c_int64, vp, vp, C.c_int]
    L.ti_painn_rollout_dlogp.argtypes = [vp, C.POINTER(RolloutDesc), vp, vp, C.c_int64, C.c_float, C.c_float, C.c_int, vp, vp, C.POINTER(C.c_int64)]
    L.ti_adw_create.restype = vp
    L.ti_adw_create.argtypes = [C.POINTER(AdwDesc), C.POINTER(C.c_double), C.c_size_t, C.c_int]
    L.ti_adw_drift.argtypes = [vp, vp, C.c_float, vp, vp, C.c_int64, vp, C.c_int]
    L.ti_adw_rollout.argtypes = [vp, C.POINTER(RolloutDesc), vp, vp, vp, C.c_int64, vp, C.POINTER(C.c_int64)]
    L.ti_adw_drift_div.argtypes = [vp, vp, C.c_float, vp, vp, C.c_int64, vp, vp, C.c_int]
    L.ti_adw_rollout_dlogp.argtypes = [vp, C.POINTER(RolloutDesc), vp, vp, vp, C.c_int64, vp, vp, C.POINTER(C.c_int64)]
    L.ti_destroy.argtypes = [vp]
    L.ti_destroy.restype = None
    L.ti_set_stream.argtypes = [vp, vp, C.c_int]
    L.ti_wait_stream.argtypes = [vp, vp]
    L.ti_painn_set_template.argtypes = [vp, C.c_int]
    L.ti_painn_template_for.argtypes = [vp, C.c_int64]
    L.ti_reserve.argtypes = [vp, C.c_int64]
    L.ti_profile_enable.argtypes = [vp, C.c_int]
    L.ti_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.ti_painn_debug_tap.argtypes = [vp, C.c_int]
    L.ti_painn_debug_read.argtypes = [vp, C.c_int, fp, C.c_size_t]
    L.ti_painn_debug_poison.argtypes = [vp, C.c_int64, C.c_float]
    L.ti_selftest.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc):
    if rc != TI_OK:
        raise TiError(rc, lib().ti_last_error().decode())


def last_error() -> str:
    return lib().ti_last_error().decode()


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def as_ptr(buf, *, shape=None, out=False, what="buffer"):
    """(void*, keepalive, is_device, device_index) for a numpy array (host), a torch tensor (cpu or cuda) or a raw int device
    address.  The kernels read and write float32, C-contiguous memory of exactly the documented shape, so anything else is
    refused here instead of being reinterpreted (tensors) or silently copied (a numpy `out` the caller would never see filled)."""
    if buf is None:
        return None, None, False, None
    if isinstance(buf, int):
        return C.c_void_p(buf), None, True, None
    if hasattr(buf, "data_ptr"):                      # torch.Tensor without importing torch
        if str(buf.dtype) != "torch.float32":
            raise TypeError(f"{what} must be float32, got {buf.dtype}")
        if not buf.is_contiguous():
            raise ValueError(f"{what} must be contiguous")
        if shape is not None and tuple(buf.shape) != tuple(shape):
            raise ValueError(f"{what} must have shape {tuple(shape)}, got {tuple(buf.shape)}")
        dev = bool(buf.is_cuda)
        return C.c_void_p(buf.data_ptr()), buf, dev, (buf.device.index if dev else None)
    if out:
        if not isinstance(buf, np.ndarray) or buf.dtype != np.float32 or not buf.flags["C_CONTIGUOUS"] or not buf.flags["WRITEABLE"]:
            raise TypeError(f"{what} must be a writable C-contiguous float32 numpy array (or a torch tensor); it is filled in place")
        a = buf
    else:
        a = np.ascontiguousarray(buf, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{what} must have shape {tuple(shape)}, got {tuple(a.shape)}")
    return C.c_void_p(a.ctypes.data), a, False, None
