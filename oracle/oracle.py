"""ctypes front-end of oracle/libti_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (see the header of
oracle/ti_oracle.c).  The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "libti_oracle.so")
_SRCS = [os.path.join(HERE, f) for f in ("ti_oracle.c", "ti_oracle_impl.h", "Makefile")] + \
        [os.path.join(HERE, "..", "include", "ti_hip.h")]


class PainnDesc(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_features", C.c_int32), ("n_layers", C.c_int32), ("n_types", C.c_int32),
                ("n_atoms", C.c_int32), ("n_edges", C.c_int32), ("temp_length", C.c_float), ("time_length", C.c_float),
                ("length_scale", C.c_float), ("temp_mean", C.c_float), ("temp_range", C.c_float), ("precision", C.c_int32)]


class AdwDesc(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("precision", C.c_int32)]


class RolloutDesc(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("n_step", C.c_int32), ("save_every", C.c_int32), ("mem", C.c_int32),
                ("eps", C.c_float), ("com_free_noise", C.c_int32), ("seed", C.c_uint64), ("traj_offset", C.c_int64),
                ("t_grid", C.POINTER(C.c_float)), ("rtol", C.c_float), ("atol", C.c_float)]


SCHEMES = {"euler": 0, "heun": 1, "em": 2}


def build(force: bool = False) -> str:
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", HERE, "-B", "libti_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.tio_painn_create.restype = C.c_void_p
        _lib.tio_adw_create.restype = C.c_void_p
        _lib.tio_normal.restype = C.c_float
        _lib.tio_normal.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32]
        _lib.tio_rollout_rows.restype = C.c_int64
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def rollout_rows(n_step: int, save_every: int) -> int:
    return int(lib().tio_rollout_rows(n_step, save_every))


def make_rollout_desc(scheme, t_grid, save_every=1, eps=0.0, seed=0, traj_offset=0, com_free_noise=0):
    t_grid = f32(t_grid)
    rd = RolloutDesc(SCHEMES[scheme] if isinstance(scheme, str) else scheme, len(t_grid), save_every, 0, eps, com_free_noise,
                     seed, traj_offset, _p(t_grid, C.c_float), 0.0, 0.0)
    rd._keep = t_grid
    return rd


class PainnOracle:
    def __init__(self, variant, F, L, A, edge_src, edge_dst, edge_type, atom_ids, flat_weights, *, n_types=25,
                 temp_length=100.0, time_length=10.0, length_scale=10.0, temperatures=(300, 400, 500, 600, 700, 800, 900, 1000)):
        temps = np.asarray(temperatures, np.float32)
        self.desc = PainnDesc(variant, F, L, n_types, A, len(edge_src), temp_length, time_length, length_scale,
                              float(temps.mean(dtype=np.float32)), float(temps.max() - temps.min()), 0)
        self.A, self.F, self.E, self.variant = A, F, len(edge_src), variant
        self.ncond = {0: 2, 1: 1, 2: 0}[variant]
        w = f32(flat_weights)
        self.h = lib().tio_painn_create(C.byref(self.desc), _p(w, C.c_float), C.c_size_t(w.size), _p(i32(edge_src), C.c_int32),
                                        _p(i32(edge_dst), C.c_int32), _p(i32(edge_type), C.c_int32), _p(i32(atom_ids), C.c_int32))
        if not self.h:
            raise RuntimeError("tio_painn_create failed (weight count / graph mismatch)")

    def drift(self, x, t, cond=None, precision=32, tap_stage=-1):
        x = f32(x)
        B = x.shape[0]
        cond = None if self.ncond == 0 else f32(cond)
        out = np.empty((B, self.A, 3), np.float32)
        taps = None
        if tap_stage >= 0:
            taps = dict(s=np.zeros((B, self.A, self.F), np.float32), v=np.zeros((B, self.A, self.F, 3), np.float32),
                        e=np.zeros((B, self.E, self.F), np.float32))
        rc = lib().tio_painn_drift(C.c_void_p(self.h), precision, _p(x, C.c_float), C.c_float(t), _p(cond, C.c_float), C.c_int64(B),
                                   _p(out, C.c_float), tap_stage, _p(taps["s"], C.c_float) if taps else None,
                                   _p(taps["v"], C.c_float) if taps else None, _p(taps["e"], C.c_float) if taps else None)
        if rc:
            raise RuntimeError(f"tio_painn_drift rc={rc}")
        return (out, taps) if taps else out

    def jvp(self, x, xdot, t, cond=None, precision=32, tap_stage=-1):
        """(b(x), (db/dx) xdot); with tap_stage >= 0 also the TANGENTS of s, v, e after that stage."""
        x, xdot = f32(x), f32(xdot)
        B = x.shape[0]
        cond = None if self.ncond == 0 else f32(cond)
        out, tan = np.empty((B, self.A, 3), np.float32), np.empty((B, self.A, 3), np.float32)
        taps = None
        if tap_stage >= 0:
            taps = dict(s=np.zeros((B, self.A, self.F), np.float32), v=np.zeros((B, self.A, self.F, 3), np.float32),
                        e=np.zeros((B, self.E, self.F), np.float32))
        rc = lib().tio_painn_jvp(C.c_void_p(self.h), precision, _p(x, C.c_float), _p(xdot, C.c_float), C.c_float(t), _p(cond, C.c_float),
                                 C.c_int64(B), _p(out, C.c_float), _p(tan, C.c_float), tap_stage,
                                 _p(taps["s"], C.c_float) if taps else None, _p(taps["v"], C.c_float) if taps else None,
                                 _p(taps["e"], C.c_float) if taps else None)
        if rc:
            raise RuntimeError(f"tio_painn_jvp rc={rc}")
        return (out, tan, taps) if taps else (out, tan)

    def drift_div(self, x, t, cond=None, precision=32):
        """(b(x) [B,A,3] f32, div [B] f64) with div = sum_ij d b_ij / d x_ij (no 1e-2 factor)."""
        x = f32(x)
        B = x.shape[0]
        cond = None if self.ncond == 0 else f32(cond)
        out, div = np.empty((B, self.A, 3), np.float32), np.empty(B, np.float64)
        rc = lib().tio_painn_drift_div(C.c_void_p(self.h), precision, _p(x, C.c_float), C.c_float(t), _p(cond, C.c_float), C.c_int64(B),
                                       _p(out, C.c_float), _p(div, C.c_double))
        if rc:
            raise RuntimeError(f"tio_painn_drift_div rc={rc}")
        return out, div

    def rollout_dlogp(self, x0, cond, t_grid, scheme="euler", save_every=1, precision=32, div_scale=1.0, reverse_ode=False):
        """(path [rows,B,A,3], dlogp [rows,B]) -- the raw second state (ambient callers multiply by 1e2)."""
        x0 = f32(x0)
        B = x0.shape[0]
        cond = None if self.ncond == 0 else f32(cond)
        rd = make_rollout_desc(scheme, t_grid, save_every)
        rows = rollout_rows(rd.n_step, save_every)
        out, dl = np.empty((rows, B, self.A, 3), np.float32), np.empty((rows, B), np.float32)
        nfe = C.c_int64(0)
        rc = lib().tio_painn_rollout_dlogp(C.c_void_p(self.h), precision, C.byref(rd), _p(x0, C.c_float), _p(cond, C.c_float), C.c_int64(B),
                                           C.c_float(div_scale), int(bool(reverse_ode)), _p(out, C.c_float), _p(dl, C.c_float), C.byref(nfe))
        if rc:
            raise RuntimeError(f"tio_painn_rollout_dlogp rc={rc}")
        return out, dl, nfe.value

    def rollout(self, x0, cond, t_grid, scheme="euler", save_every=1, precision=32, **kw):
        x0 = f32(x0)
        B = x0.shape[0]
        cond = None if self.ncond == 0 else f32(cond)
        rd = make_rollout_desc(scheme, t_grid, save_every, **kw)
        out = np.empty((rollout_rows(rd.n_step, save_every), B, self.A, 3), np.float32)
        nfe = C.c_int64(0)
        rc = lib().tio_painn_rollout(C.c_void_p(self.h), precision, C.byref(rd), _p(x0, C.c_float), _p(cond, C.c_float), C.c_int64(B),
                                     _p(out, C.c_float), C.byref(nfe))
        if rc:
            raise RuntimeError(f"tio_painn_rollout rc={rc}")
        return out, nfe.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().tio_painn_destroy(C.c_void_p(self.h))
            self.h = None


class AdwOracle:
    def __init__(self, hidden, num_layers, flat_weights_f64):
        self.desc = AdwDesc(hidden, num_layers, 0)
        w = np.ascontiguousarray(flat_weights_f64, np.float64)
        self.h = lib().tio_adw_create(C.byref(self.desc), _p(w, C.c_double), C.c_size_t(w.size))
        if not self.h:
            raise RuntimeError("tio_adw_create failed (weight count mismatch)")

    def drift(self, x, t, beta0, beta1, precision=64):
        ty, ct = (np.float64, C.c_double) if precision == 64 else (np.float32, C.c_float)
        x, b0, b1 = (np.ascontiguousarray(np.broadcast_to(a, np.shape(x)), ty) for a in (x, beta0, beta1))
        out = np.empty_like(x)
        fn = lib().tio_adw_drift_f64 if precision == 64 else lib().tio_adw_drift_f32
        fn(C.c_void_p(self.h), _p(x, ct), ct(t), _p(b0, ct), _p(b1, ct), C.c_int64(x.size), _p(out, ct))
        return out

    def drift_div(self, x, t, beta0, beta1):
        """(b, d b / d x) in fp64; the reference's compute_divergence is d b / d x * 1e-2 (ode_wrapper.py:67)."""
        x, b0, b1 = (np.ascontiguousarray(np.broadcast_to(a, np.shape(x)), np.float64) for a in (x, beta0, beta1))
        out, div = np.empty_like(x), np.empty_like(x)
        lib().tio_adw_drift_div_f64(C.c_void_p(self.h), _p(x, C.c_double), C.c_double(t), _p(b0, C.c_double), _p(b1, C.c_double),
                                    C.c_int64(x.size), _p(out, C.c_double), _p(div, C.c_double))
        return out, div

    def rollout(self, x0, beta0, beta1, t_grid, scheme="euler", save_every=1, return_dlogp=False, **kw):
        x0, b0, b1 = (np.ascontiguousarray(np.broadcast_to(a, np.shape(x0)), np.float64) for a in (x0, beta0, beta1))
        rd = make_rollout_desc(scheme, t_grid, save_every, **kw)
        out = np.empty((rollout_rows(rd.n_step, save_every), x0.size), np.float64)
        dl = np.empty_like(out) if return_dlogp else None
        nfe = C.c_int64(0)
        lib().tio_adw_rollout_f64(C.c_void_p(self.h), C.byref(rd), _p(x0, C.c_double), _p(b0, C.c_double), _p(b1, C.c_double),
                                  C.c_int64(x0.size), _p(out, C.c_double), _p(dl, C.c_double), C.byref(nfe))
        return (out, dl, nfe.value) if return_dlogp else (out, nfe.value)


def normal(seed, traj, step, comp) -> float:
    return float(lib().tio_normal(seed, traj, step, comp))


def num_threads() -> int:
    return int(lib().tio_num_threads())
