"""Thin Python objects over the C ABI handles (include/ti_hip.h).  No arithmetic happens here.

Buffers may be numpy arrays (host memory, staged by the library) or CUDA/HIP torch tensors (used in place through
``data_ptr()``; nothing from torch is imported in this module).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import weights as W


def time_grid(start: float, end: float, n_step: int) -> np.ndarray:
    """The reference grid ``torch.linspace(start, end, n_step)`` as float32
    (/root/reference/mdqm9/thermo/ambient/integrators.py:43).  PyTorch is used for it when importable, so the values are
    bit-identical to the reference's; otherwise the same two-sided formula is evaluated in numpy (within 1 ulp: torch's
    vectorised kernel rounds ``start + step * i`` in two stages)."""
    try:
        import torch
        return torch.linspace(float(start), float(end), int(n_step), dtype=torch.float32).numpy().copy()
    except ImportError:
        return _time_grid_numpy(start, end, n_step)


def _time_grid_numpy(start: float, end: float, n_step: int) -> np.ndarray:
    start, end = np.float32(start), np.float32(end)
    if n_step == 1:
        return np.asarray([start], np.float32)
    step = np.float32((end - start) / np.float32(n_step - 1))
    i = np.arange(n_step, dtype=np.int64)
    lo = start + step * i.astype(np.float32)
    hi = end - step * (n_step - 1 - i).astype(np.float32)
    return np.where(i < n_step // 2, lo, hi).astype(np.float32)


def _rollout_desc(scheme, t_grid, save_every, mem, eps, seed, traj_offset, com_free_noise, rtol=0.0, atol=0.0):
    t_grid = np.ascontiguousarray(t_grid, np.float32)
    if t_grid.ndim != 1 or t_grid.size < 1:
        raise ValueError("t_grid must be a non-empty 1-D array")
    if isinstance(scheme, str):
        if scheme not in _lib.SCHEMES:
            raise ValueError(f"unknown scheme {scheme!r}; expected one of {sorted(_lib.SCHEMES)}")
        scheme = _lib.SCHEMES[scheme]
    rd = _lib.RolloutDesc(scheme, t_grid.size, int(save_every), mem, float(eps), int(bool(com_free_noise)), int(seed),
                          int(traj_offset), _lib.fptr(t_grid), float(rtol), float(atol))
    rd._keep = t_grid
    return rd


def _alloc_like(template, shape):
    """Output buffer living where `template` lives (numpy -> numpy, cuda tensor -> cuda tensor)."""
    if hasattr(template, "data_ptr"):
        return template.new_empty(shape)
    return np.empty(shape, np.float32)


class _Engine:
    h = None

    def close(self):
        if self.h:
            _lib.lib().ti_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int | None):
        _lib.check(_lib.lib().ti_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def reserve(self, B: int):
        _lib.check(_lib.lib().ti_reserve(self.h, int(B)))

    def profile(self, on: bool = True):
        _lib.check(_lib.lib().ti_profile_enable(self.h, int(on)))

    def profile_read(self, kernel: str):
        n, ms = C.c_int64(0), C.c_double(0.0)
        _lib.check(_lib.lib().ti_profile_read(self.h, _lib.KERNELS[kernel], C.byref(n), C.byref(ms)))
        return n.value, ms.value


class PainnEngine(_Engine):
    """cPaiNN drift + fixed-step integrator for one molecular species (homogeneous batches)."""

    def __init__(self, variant, F, L, A, edge_src, edge_dst, edge_type, atom_ids, flat_weights, *, n_types=25, temp_length=10.0,
                 time_length=10.0, length_scale=10.0, temperatures=(300, 400, 500, 600, 700, 800, 900, 1000), device=0, precision="f32"):
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self.precision = precision
        temps = np.asarray(temperatures, np.float32)
        es, ed, et, ai = (np.ascontiguousarray(a, np.int32) for a in (edge_src, edge_dst, edge_type, atom_ids))
        if not (es.shape == ed.shape == et.shape) or es.ndim != 1 or ai.shape != (A,):
            raise ValueError("edge_src/edge_dst/edge_type must be 1-D and equally long; atom_ids must have A entries")
        self.variant, self.F, self.L, self.A, self.E = int(variant), int(F), int(L), int(A), int(es.size)
        self.ncond = W.N_COND[self.variant]
        self.desc = _lib.PainnDesc(self.variant, self.F, self.L, int(n_types), self.A, self.E, float(temp_length), float(time_length),
                                   float(length_scale), float(temps.mean(dtype=np.float32)), float(temps.max() - temps.min()), _lib.PRECISIONS[precision])
        w = np.ascontiguousarray(flat_weights, np.float32)
        self.device = int(device)
        self.h = _lib.lib().ti_painn_create(C.byref(self.desc), _lib.fptr(w), w.size, _lib.iptr(es), _lib.iptr(ed), _lib.iptr(et),
                                            _lib.iptr(ai), self.device)
        if not self.h:
            raise _lib.TiError(-1, _lib.last_error())

    def _bufs(self, x, cond):
        xp, xk, xdev = _lib.as_ptr(x)
        cp, ck, cdev = _lib.as_ptr(cond if self.ncond else None)
        if self.ncond and cond is None:
            raise ValueError("this variant needs per-node conditioning (cond)")
        if self.ncond and cdev != xdev:
            raise ValueError("x and cond must live in the same memory space")
        return xp, cp, xdev, (xk, ck)

    def drift(self, x, t, cond=None, out=None):
        """x [B,A,3] -> drift [B,A,3] at time t."""
        B = int(x.shape[0])
        if tuple(x.shape[1:]) != (self.A, 3):
            raise ValueError(f"x must be [B,{self.A},3]")
        xp, cp, dev, keep = self._bufs(x, cond)
        out = _alloc_like(x if dev else None, (B, self.A, 3)) if out is None else out
        op, _, odev = _lib.as_ptr(out)
        if odev != dev:
            raise ValueError("out must live where x lives")
        _lib.check(_lib.lib().ti_painn_drift(self.h, xp, float(t), cp, B, op, _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out

    def rollout(self, x0, cond, t_grid, scheme="euler", save_every=1, eps=0.0, seed=0, traj_offset=0, com_free_noise=False, out=None,
                rtol=1e-4, atol=1e-4):
        """Returns (path [rows,B,A,3], n_fevals).  scheme: 'euler' | 'heun' | 'em' | 'midpoint' | 'rk4' on the grid, or 'dopri5'
        (adaptive, tolerances rtol / atol; the grid then only selects the output times)."""
        B = int(x0.shape[0])
        if tuple(x0.shape[1:]) != (self.A, 3):
            raise ValueError(f"x0 must be [B,{self.A},3]")
        xp, cp, dev, keep = self._bufs(x0, cond)
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if dev else _lib.MEM_HOST, eps, seed, traj_offset, com_free_noise, rtol, atol)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        out = _alloc_like(x0 if dev else None, (rows, B, self.A, 3)) if out is None else out
        op, _, odev = _lib.as_ptr(out)
        if odev != dev:
            raise ValueError("out must live where x0 lives")
        nfe = C.c_int64(0)
        _lib.check(_lib.lib().ti_painn_rollout(self.h, C.byref(rd), xp, cp, B, op, C.byref(nfe)))
        return out, nfe.value

    # ---- forward-mode derivative, exact divergence, dlogp
    def jvp(self, x, xdot, t, cond=None):
        """(b(x), (d b / d x) xdot), both [B,A,3]."""
        B = int(x.shape[0])
        if tuple(x.shape[1:]) != (self.A, 3) or tuple(xdot.shape) != tuple(x.shape):
            raise ValueError(f"x and xdot must be [B,{self.A},3]")
        xp, cp, dev, keep = self._bufs(x, cond)
        tp, tk, tdev = _lib.as_ptr(xdot)
        if tdev != dev:
            raise ValueError("xdot must live where x lives")
        out, tan = _alloc_like(x if dev else None, (B, self.A, 3)), _alloc_like(x if dev else None, (B, self.A, 3))
        _lib.check(_lib.lib().ti_painn_drift_jvp(self.h, xp, tp, float(t), cp, B, _lib.as_ptr(out)[0], _lib.as_ptr(tan)[0],
                                                 _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out, tan

    def drift_div(self, x, t, cond=None):
        """(b(x) [B,A,3], div [B]) with div = sum_ij d b_ij / d x_ij -- the reference's compute_divergence without its 1e-2."""
        B = int(x.shape[0])
        if tuple(x.shape[1:]) != (self.A, 3):
            raise ValueError(f"x must be [B,{self.A},3]")
        xp, cp, dev, keep = self._bufs(x, cond)
        out, div = _alloc_like(x if dev else None, (B, self.A, 3)), _alloc_like(x if dev else None, (B,))
        _lib.check(_lib.lib().ti_painn_drift_div(self.h, xp, float(t), cp, B, _lib.as_ptr(out)[0], _lib.as_ptr(div)[0],
                                                 _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out, div

    def rollout_dlogp(self, x0, cond, t_grid, scheme="euler", save_every=1, div_scale=1.0, out_scale=1.0, reverse_ode=False,
                      rtol=1e-4, atol=1e-4):
        """Two-state rollout (x, dlogp): returns (path [rows,B,A,3], dlogp [rows,B], n_fevals).  d(dlogp)/dt = -div_scale * div
        (reverse_ode: (-b, +div_scale * div) on the descending grid the caller passes), dlogp is written * out_scale."""
        B = int(x0.shape[0])
        if tuple(x0.shape[1:]) != (self.A, 3):
            raise ValueError(f"x0 must be [B,{self.A},3]")
        xp, cp, dev, keep = self._bufs(x0, cond)
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if dev else _lib.MEM_HOST, 0.0, 0, 0, False, rtol, atol)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        out, dl = _alloc_like(x0 if dev else None, (rows, B, self.A, 3)), _alloc_like(x0 if dev else None, (rows, B))
        nfe = C.c_int64(0)
        _lib.check(_lib.lib().ti_painn_rollout_dlogp(self.h, C.byref(rd), xp, cp, B, float(div_scale), float(out_scale), int(bool(reverse_ode)),
                                                     _lib.as_ptr(out)[0], _lib.as_ptr(dl)[0], C.byref(nfe)))
        return out, dl, nfe.value

    # ---- parity-test taps
    def debug_tap(self, stage: int):
        _lib.check(_lib.lib().ti_painn_debug_tap(self.h, int(stage)))

    def debug_read(self, what: str, B: int):
        """what: 's' | 'v' | 'e', or 'ts' | 'tv' | 'te' for the tangents of the last jvp() call."""
        shape = {"s": (B, self.A, self.F), "v": (B, self.A, 3, self.F), "e": (B, self.E, self.F)}[what[-1]]
        out = np.empty(shape, np.float32)
        _lib.check(_lib.lib().ti_painn_debug_read(self.h, {"s": 0, "v": 1, "e": 2, "ts": 3, "tv": 4, "te": 5}[what], _lib.fptr(out), out.size))
        return out


class AdwEngine(_Engine):
    """FCNetMultiBeta drift (+ exact divergence) and fixed-step integrator for the 1-D double well."""

    def __init__(self, hidden, num_layers, flat_weights_f64, device=0, precision="f32"):
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self.hidden, self.num_layers, self.precision = int(hidden), int(num_layers), precision
        self.desc = _lib.AdwDesc(self.hidden, self.num_layers, _lib.PRECISIONS[precision])
        w = np.ascontiguousarray(flat_weights_f64, np.float64)
        self.device = int(device)
        self.h = _lib.lib().ti_adw_create(C.byref(self.desc), w.ctypes.data_as(C.POINTER(C.c_double)), w.size, self.device)
        if not self.h:
            raise _lib.TiError(-1, _lib.last_error())

    @staticmethod
    def _same_space(*bufs):
        flags = {bool(_lib.as_ptr(b)[2]) for b in bufs}
        if len(flags) != 1:
            raise ValueError("x, beta0 and beta1 must live in the same memory space")
        return flags.pop()

    def drift(self, x, t, beta0, beta1, out=None, return_div=False):
        """b(x, t) [B]; with return_div also d b / d x (the 1-D divergence, reference scaling NOT applied)."""
        B = int(x.shape[0])
        dev = self._same_space(x, beta0, beta1)
        (xp, xk, _), (b0p, b0k, _), (b1p, b1k, _) = _lib.as_ptr(x), _lib.as_ptr(beta0), _lib.as_ptr(beta1)
        out = _alloc_like(x if dev else None, (B,)) if out is None else out
        op, _, _ = _lib.as_ptr(out)
        mem = _lib.MEM_DEVICE if dev else _lib.MEM_HOST
        if not return_div:
            _lib.check(_lib.lib().ti_adw_drift(self.h, xp, float(t), b0p, b1p, B, op, mem))
            return out
        div = _alloc_like(x if dev else None, (B,))
        dp, _, _ = _lib.as_ptr(div)
        _lib.check(_lib.lib().ti_adw_drift_div(self.h, xp, float(t), b0p, b1p, B, op, dp, mem))
        return out, div

    def rollout(self, x0, beta0, beta1, t_grid, scheme="euler", save_every=1, eps=0.0, seed=0, traj_offset=0, out=None,
                return_dlogp=False, rtol=1e-4, atol=1e-4):
        """(path [rows,B], n_fevals), or (path, dlogp [rows,B] (already * 1e2 like the reference), n_fevals)."""
        B = int(x0.shape[0])
        dev = self._same_space(x0, beta0, beta1)
        (xp, xk, _), (b0p, b0k, _), (b1p, b1k, _) = _lib.as_ptr(x0), _lib.as_ptr(beta0), _lib.as_ptr(beta1)
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if dev else _lib.MEM_HOST, eps, seed, traj_offset, False, rtol, atol)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        out = _alloc_like(x0 if dev else None, (rows, B)) if out is None else out
        op, _, _ = _lib.as_ptr(out)
        nfe = C.c_int64(0)
        if not return_dlogp:
            _lib.check(_lib.lib().ti_adw_rollout(self.h, C.byref(rd), xp, b0p, b1p, B, op, C.byref(nfe)))
            return out, nfe.value
        dl = _alloc_like(x0 if dev else None, (rows, B))
        dp, _, _ = _lib.as_ptr(dl)
        _lib.check(_lib.lib().ti_adw_rollout_dlogp(self.h, C.byref(rd), xp, b0p, b1p, B, op, dp, C.byref(nfe)))
        return out, dl, nfe.value


def selftest(device: int = 0):
    _lib.check(_lib.lib().ti_selftest(int(device)))
