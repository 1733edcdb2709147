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


def _rollout_desc(scheme, t_grid, save_every, mem, eps, seed, traj_offset, com_free_noise, rtol=0.0, atol=0.0, step_offset=0):
    t_grid = np.ascontiguousarray(t_grid, np.float32)
    if t_grid.ndim != 1 or t_grid.size < 1:
        raise ValueError("t_grid must be a non-empty 1-D array")
    if isinstance(scheme, str):
        if scheme not in _lib.SCHEMES:
            raise ValueError(f"unknown scheme {scheme!r}; expected one of {sorted(_lib.SCHEMES)}")
        scheme = _lib.SCHEMES[scheme]
    rd = _lib.RolloutDesc(scheme, t_grid.size, int(save_every), mem, float(eps), int(bool(com_free_noise)), int(seed),
                          int(traj_offset), _lib.fptr(t_grid), float(rtol), float(atol), int(step_offset))
    rd._keep = t_grid
    return rd


def _alloc_like(template, shape):
    """Output buffer living where `template` lives (numpy -> numpy, cuda tensor -> cuda tensor)."""
    if hasattr(template, "data_ptr"):
        return template.new_empty(shape)
    return np.empty(shape, np.float32)


class _Engine:
    h = None
    device = 0

    def close(self):
        if self.h:
            _lib.lib().ti_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int | None, external: bool = True):
        """Run on the caller's HIP stream (`hip_stream` = hipStream_t as int; 0 / None = the null stream, which is what torch's
        default stream reports), or with external=False on the handle's own stream again."""
        _lib.check(_lib.lib().ti_set_stream(self.h, C.c_void_p(hip_stream or 0), 1 if external else 0))

    def wait_stream(self, hip_stream: int | None):
        """Order the handle's stream after everything enqueued so far on `hip_stream` (0 / None = the null stream)."""
        _lib.check(_lib.lib().ti_wait_stream(self.h, C.c_void_p(hip_stream or 0)))

    def _ptrs(self, *specs):
        """as_ptr over (buffer, shape, is_output, name) specs; all present buffers must live in one memory space.  Device buffers
        must be on this engine's GPU, and the handle's stream is ordered after torch's current stream on that GPU first, so a
        tensor produced by a still-running torch kernel is never read early.  Returns (pointers, is_device, keepalives)."""
        ptrs, keep, spaces, devs = [], [], set(), set()
        for buf, shape, out, what in specs:
            pt, k, dev, idx = _lib.as_ptr(buf, shape=shape, out=out, what=what)
            ptrs.append(pt); keep.append(k)
            if buf is not None:
                spaces.add(dev)
                if idx is not None:
                    devs.add(idx)
        if len(spaces) > 1:
            raise ValueError("all buffers of a call must live in the same memory space (all host or all on the GPU)")
        dev = bool(spaces and spaces.pop())
        if dev:
            if devs - {self.device}:
                raise ValueError(f"tensors live on cuda:{sorted(devs)} but this engine was created on device {self.device}")
            if devs:                                  # torch tensors (raw int addresses carry no stream: the caller orders them)
                import torch
                self.wait_stream(torch.cuda.current_stream(self.device).cuda_stream)
        return ptrs, dev, keep

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

    TEMPLATES = {"auto": -1, "throughput": 0, "latency": 1, "pair": 2}

    def set_template(self, which: str = "auto"):
        """Pin the edge-row layout ('throughput' | 'latency': directed rows; 'pair': pair-major rows, the filter branch once per atom
        pair -- falls back to 'throughput' where no pair layout exists) or let each call choose from its batch size ('auto')."""
        _lib.check(_lib.lib().ti_painn_set_template(self.h, self.TEMPLATES[which]))

    def template_for(self, B: int) -> str:
        """The layout a drift / rollout call over B molecules would use."""
        return {0: "throughput", 1: "latency", 2: "pair"}[_lib.lib().ti_painn_template_for(self.h, int(B))]

    def _check_x(self, x, name="x"):
        if x is None or len(x.shape) != 3 or tuple(x.shape[1:]) != (self.A, 3):
            raise ValueError(f"{name} must be [B,{self.A},3]")
        return int(x.shape[0])

    def _cond_spec(self, cond, B):
        if self.ncond and cond is None:
            raise ValueError("this variant needs per-node conditioning (cond)")
        return (cond if self.ncond else None, (B, self.A, self.ncond), False, "cond")

    def drift(self, x, t, cond=None, out=None):
        """x [B,A,3] -> drift [B,A,3] at time t."""
        B = self._check_x(x)
        if out is None:
            out = _alloc_like(x if hasattr(x, "data_ptr") and x.is_cuda else None, (B, self.A, 3))
        (xp, cp, op), dev, keep = self._ptrs((x, (B, self.A, 3), False, "x"), self._cond_spec(cond, B), (out, (B, self.A, 3), True, "out"))
        _lib.check(_lib.lib().ti_painn_drift(self.h, xp, float(t), cp, B, op, _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out

    def rollout(self, x0, cond, t_grid, scheme="euler", save_every=1, eps=0.0, seed=0, traj_offset=0, com_free_noise=False, out=None,
                rtol=1e-4, atol=1e-4, step_offset=0):
        """Returns (path [rows,B,A,3], n_fevals).  scheme: 'euler' | 'heun' | 'em' | 'midpoint' | 'rk4' on the grid, or 'dopri5'
        (adaptive, tolerances rtol / atol; the grid then only selects the output times).  step_offset: EM noise counter of the
        call's first step (pass the number of steps already taken when continuing a trajectory)."""
        B = self._check_x(x0, "x0")
        on_gpu = hasattr(x0, "data_ptr") and x0.is_cuda
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if on_gpu else _lib.MEM_HOST, eps, seed, traj_offset, com_free_noise,
                           rtol, atol, step_offset)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        if out is None:
            out = _alloc_like(x0 if on_gpu else None, (rows, B, self.A, 3))
        (xp, cp, op), dev, keep = self._ptrs((x0, (B, self.A, 3), False, "x0"), self._cond_spec(cond, B), (out, (rows, B, self.A, 3), True, "out"))
        nfe = C.c_int64(0)
        _lib.check(_lib.lib().ti_painn_rollout(self.h, C.byref(rd), xp, cp, B, op, C.byref(nfe)))
        return out, nfe.value

    # ---- forward-mode derivative, exact divergence, dlogp
    def jvp(self, x, xdot, t, cond=None):
        """(b(x), (d b / d x) xdot), both [B,A,3]."""
        B = self._check_x(x)
        like = x if hasattr(x, "data_ptr") and x.is_cuda else None
        out, tan = _alloc_like(like, (B, self.A, 3)), _alloc_like(like, (B, self.A, 3))
        (xp, tp, cp, op, tnp), dev, keep = self._ptrs((x, (B, self.A, 3), False, "x"), (xdot, (B, self.A, 3), False, "xdot"), self._cond_spec(cond, B),
                                                      (out, None, True, "out"), (tan, None, True, "out_tan"))
        _lib.check(_lib.lib().ti_painn_drift_jvp(self.h, xp, tp, float(t), cp, B, op, tnp, _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out, tan

    def drift_div(self, x, t, cond=None):
        """(b(x) [B,A,3], div [B]) with div = sum_ij d b_ij / d x_ij -- the reference's compute_divergence without its 1e-2."""
        B = self._check_x(x)
        like = x if hasattr(x, "data_ptr") and x.is_cuda else None
        out, div = _alloc_like(like, (B, self.A, 3)), _alloc_like(like, (B,))
        (xp, cp, op, dp), dev, keep = self._ptrs((x, (B, self.A, 3), False, "x"), self._cond_spec(cond, B), (out, None, True, "out"), (div, None, True, "out_div"))
        _lib.check(_lib.lib().ti_painn_drift_div(self.h, xp, float(t), cp, B, op, dp, _lib.MEM_DEVICE if dev else _lib.MEM_HOST))
        return out, div

    def rollout_dlogp(self, x0, cond, t_grid, scheme="euler", save_every=1, div_scale=1.0, out_scale=1.0, reverse_ode=False,
                      rtol=1e-4, atol=1e-4):
        """Two-state rollout (x, dlogp): returns (path [rows,B,A,3], dlogp [rows,B], n_fevals).  d(dlogp)/dt = -div_scale * div
        (reverse_ode: (-b, +div_scale * div) on the descending grid the caller passes), dlogp is written * out_scale."""
        B = self._check_x(x0, "x0")
        on_gpu = hasattr(x0, "data_ptr") and x0.is_cuda
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if on_gpu else _lib.MEM_HOST, 0.0, 0, 0, False, rtol, atol)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        out, dl = _alloc_like(x0 if on_gpu else None, (rows, B, self.A, 3)), _alloc_like(x0 if on_gpu else None, (rows, B))
        (xp, cp, op, dp), dev, keep = self._ptrs((x0, (B, self.A, 3), False, "x0"), self._cond_spec(cond, B), (out, None, True, "out"), (dl, None, True, "out_dlogp"))
        nfe = C.c_int64(0)
        _lib.check(_lib.lib().ti_painn_rollout_dlogp(self.h, C.byref(rd), xp, cp, B, float(div_scale), float(out_scale), int(bool(reverse_ode)),
                                                     op, dp, C.byref(nfe)))
        return out, dl, nfe.value

    # ---- parity-test taps
    def debug_tap(self, stage: int):
        _lib.check(_lib.lib().ti_painn_debug_tap(self.h, int(stage)))

    def debug_poison(self, B: int, value: float):
        """Test hook: fill the per-atom accumulators of the workspace for B molecules with `value` (first-touch test)."""
        _lib.check(_lib.lib().ti_painn_debug_poison(self.h, int(B), float(value)))

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

    def drift(self, x, t, beta0, beta1, out=None, return_div=False):
        """b(x, t) [B]; with return_div also d b / d x (the 1-D divergence, reference scaling NOT applied)."""
        B = int(x.shape[0])
        like = x if hasattr(x, "data_ptr") and x.is_cuda else None
        if out is None:
            out = _alloc_like(like, (B,))
        div = _alloc_like(like, (B,)) if return_div else None
        (xp, b0p, b1p, op, dp), dev, keep = self._ptrs((x, (B,), False, "x"), (beta0, (B,), False, "beta0"), (beta1, (B,), False, "beta1"),
                                                       (out, (B,), True, "out"), (div, None, True, "out_div"))
        mem = _lib.MEM_DEVICE if dev else _lib.MEM_HOST
        if not return_div:
            _lib.check(_lib.lib().ti_adw_drift(self.h, xp, float(t), b0p, b1p, B, op, mem))
            return out
        _lib.check(_lib.lib().ti_adw_drift_div(self.h, xp, float(t), b0p, b1p, B, op, dp, mem))
        return out, div

    def rollout(self, x0, beta0, beta1, t_grid, scheme="euler", save_every=1, eps=0.0, seed=0, traj_offset=0, out=None,
                return_dlogp=False, rtol=1e-4, atol=1e-4, step_offset=0):
        """(path [rows,B], n_fevals), or (path, dlogp [rows,B] (already * 1e2 like the reference), n_fevals)."""
        B = int(x0.shape[0])
        on_gpu = hasattr(x0, "data_ptr") and x0.is_cuda
        rd = _rollout_desc(scheme, t_grid, save_every, _lib.MEM_DEVICE if on_gpu else _lib.MEM_HOST, eps, seed, traj_offset, False, rtol, atol,
                           step_offset)
        rows = int(_lib.lib().ti_rollout_rows(rd.n_step, rd.save_every))
        if out is None:
            out = _alloc_like(x0 if on_gpu else None, (rows, B))
        dl = _alloc_like(x0 if on_gpu else None, (rows, B)) if return_dlogp else None
        (xp, b0p, b1p, op, dp), dev, keep = self._ptrs((x0, (B,), False, "x0"), (beta0, (B,), False, "beta0"), (beta1, (B,), False, "beta1"),
                                                       (out, (rows, B), True, "out"), (dl, None, True, "out_dlogp"))
        nfe = C.c_int64(0)
        if not return_dlogp:
            _lib.check(_lib.lib().ti_adw_rollout(self.h, C.byref(rd), xp, b0p, b1p, B, op, C.byref(nfe)))
            return out, nfe.value
        _lib.check(_lib.lib().ti_adw_rollout_dlogp(self.h, C.byref(rd), xp, b0p, b1p, B, op, dp, C.byref(nfe)))
        return out, dl, nfe.value


def selftest(device: int = 0):
    _lib.check(_lib.lib().ti_selftest(int(device)))
