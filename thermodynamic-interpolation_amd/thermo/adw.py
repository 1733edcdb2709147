"""Drop-in mirror of the reference adw sampling API on top of libti_hip.so.

  FCNetMultiBeta      <- /root/reference/adw/thermo/models/simple.py:5-41
  ODEWrapper          <- /root/reference/adw/thermo/models/ode_wrapper.py:11-68   (drift and exact divergence)
  StandardIntegrator  <- /root/reference/adw/thermo/integrators.py:11-68

Same class names, constructor arguments and return shapes; tensors in, tensors out (numpy also accepted).  The modules hold
weights only -- all arithmetic runs in the HIP library (fp32 on the device; the reference nets are fp64, adw/train.py:29).
"""
from __future__ import annotations

import numpy as np

from .. import engine as _engine
from .. import synthetic as _syn
from .. import weights as _W
from . import _common as C


class FCNetMultiBeta:
    """Weights-only shell with the reference constructor signature and state_dict key layout."""

    def __init__(self, in_size, out_size, hidden_size, num_layers):
        if in_size != 1 or out_size != 1:
            raise NotImplementedError("the HIP path covers the reference's 1-D double well (in_size = out_size = 1)")
        self.in_size, self.out_size, self.hidden_size, self.num_layers = in_size, out_size, hidden_size, num_layers
        self._spec = _W.adw_param_spec(hidden_size, num_layers)
        # the reference initialises with torch's default Linear init; weights normally arrive via load_state_dict / torch.load
        self._sd = _syn.make_state_dict(self._spec, seed=0, dtype=np.float64)
        self._engine, self._device = None, 0
        self.precision = "f32"                  # 'f16x2': split-fp16 matrix path; set before first use
        self.training = False

    # -- torch.nn.Module surface used by the sampling driver (adw/sample.py:39, :84-88)
    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, state_dict, strict=True):
        flat = _W.flatten_state_dict(state_dict, self._spec, dtype=np.float64, strict=strict)
        self._sd = _W.unflatten(flat, self._spec)
        self._engine = None
        return self

    @classmethod
    def from_torch_module(cls, module):
        """Build from a reference ``FCNetMultiBeta`` instance (what ``torch.load(config.sampling_model)`` returns)."""
        sd = module.state_dict()
        hidden = int(sd["net.0.weight"].shape[0])
        n_lin = sum(1 for k in sd if k.startswith("net.") and k.endswith(".weight"))
        return cls(1, 1, hidden, n_lin - 1).load_state_dict(sd)

    def eval(self):
        self.training = False
        return self

    def double(self):
        return self

    def float(self):
        return self

    def to(self, device=None, *a, **k):
        idx = getattr(device, "index", None)
        if isinstance(device, int):
            idx = device
        elif isinstance(device, str) and ":" in device:
            idx = int(device.split(":")[1])
        if idx is not None and idx != self._device:
            self._device, self._engine = idx, None
        return self

    def parameters(self):
        return iter(self._sd.values())

    def engine(self) -> _engine.AdwEngine:
        if self._engine is None:
            flat = _W.flatten_state_dict(self._sd, self._spec, dtype=np.float64)
            self._engine = _engine.AdwEngine(self.hidden_size, self.num_layers, flat, device=self._device, precision=self.precision)
        return self._engine

    def forward(self, x0s, xts, ts, beta0s, beta1s):
        """net([xts, ts, beta_embed([beta0s, beta1s, ts])]) -> [B, 1].  ``x0s`` is unused, as in the reference (simple.py:38-41).
        ``ts`` must hold one value (the sampler always passes ones_like(x) * t, ode_wrapper.py:47)."""
        t = C.to_numpy(ts, np.float64).ravel()
        if t.size and np.ptp(t) != 0.0:
            raise NotImplementedError("per-row times are a training-only input; the sampling path evaluates one t per call")
        x = np.ascontiguousarray(C.to_numpy(xts, np.float32).reshape(-1))
        b0 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta0s, np.float32).reshape(-1), x.shape))
        b1 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta1s, np.float32).reshape(-1), x.shape))
        out = self.engine().drift(x, float(t[0]) if t.size else 0.0, b0, b1)
        return C.like(out.reshape(-1, 1), xts)

    __call__ = forward


class ODEWrapper:
    """forward(t, states, x0s, beta0s, beta1s) -> b, or (b, -divergence) with return_dlogp
    (divergence = d b / d x * 1e-2 like ode_wrapper.py:55-67; reverse_ode flips both signs)."""

    def __init__(self, b, return_dlogp=False, reverse_ode=False):
        self.b, self.return_dlogp, self.reverse_ode = b, return_dlogp, reverse_ode

    def forward(self, integration_time, states, x0s, beta0s, beta1s):
        xs = states[0] if isinstance(states, (tuple, list)) else states
        t = float(integration_time)
        if not self.return_dlogp:
            ts = np.full(C.to_numpy(xs).shape, t, np.float32)
            return self.b.forward(x0s, xs, ts, beta0s, beta1s)
        x = np.ascontiguousarray(C.to_numpy(xs, np.float32).reshape(-1))
        b0 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta0s, np.float32).reshape(-1), x.shape))
        b1 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta1s, np.float32).reshape(-1), x.shape))
        b, div = self.b.engine().drift(x, t, b0, b1, return_div=True)
        b, div = C.like(b.reshape(-1, 1), xs), C.like(div * np.float32(1e-2), xs)
        return (b, -div) if not self.reverse_ode else (-b, div)

    __call__ = forward


class StandardIntegrator:
    """rollout(x0s, beta0s, beta1s) -> (x [n_saved, B, 1], dlogp)   -- reference: (x [n_step, B, 1], dlogp * 1e2).

    ``method``: 'dopri5' (adaptive, rtol / atol; the grid selects the output times) or 'euler' | 'midpoint' | 'rk4' | 'heun' | 'em'
    on the grid torch.linspace(start, end, n_step) (n_step - 1 steps).  Extra keyword
    arguments are build-defined: ``eps``/``seed`` (EM noise), ``save_every`` (1 keeps every grid point like the reference;
    0 keeps the end state only).  With return_dlogp=True the second ODE state d(dlogp)/dt = -div * 1e-2 is integrated with the
    same scheme and returned * 1e2 as [n_saved, B, 1], like the reference (integrators.py:38-68).  With return_dlogp=False the
    reference evaluates ``None * 1e2`` and raises (integrators.py:42,68); here dlogp is returned as None instead.
    """

    def __init__(self, b, method: str = "dopri5", n_step: int = 100, atol: float = 1e-4, rtol: float = 1e-4, start: float = 0.0,
                 end: float = 1.0, return_dlogp=False, *, eps: float = 0.0, seed: int = 0, save_every: int = 1):
        self.method = C.check_method(method)
        self.ode_wrapper = ODEWrapper(b, return_dlogp=return_dlogp)
        self.start, self.end, self.rtol, self.atol = start, end, rtol, atol
        self.n_step, self.return_dlogp = n_step, return_dlogp
        self.eps, self.seed, self.save_every = eps, seed, save_every

    def rollout(self, x0s, beta0s, beta1s, traj_offset: int = 0):
        if len(x0s.shape) != 2 or x0s.shape[1] != 1:
            raise ValueError("x0s must be [batch, 1]")
        B = int(x0s.shape[0])
        if C.is_cuda(x0s):                                   # stay in HBM: data_ptr() in, CUDA tensors out
            import torch
            x0 = x0s.detach().to(torch.float32).reshape(B).contiguous()
            b0 = torch.as_tensor(beta0s, device=x0.device).to(torch.float32).reshape(-1).expand(B).contiguous()
            b1 = torch.as_tensor(beta1s, device=x0.device).to(torch.float32).reshape(-1).expand(B).contiguous()
        else:
            x0 = np.ascontiguousarray(C.to_numpy(x0s, np.float32)[:, 0])
            b0 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta0s, np.float32).reshape(-1), (B,)))
            b1 = np.ascontiguousarray(np.broadcast_to(C.to_numpy(beta1s, np.float32).reshape(-1), (B,)))
        grid = _engine.time_grid(self.start, self.end, self.n_step)
        res = self.ode_wrapper.b.engine().rollout(x0, b0, b1, grid, scheme=self.method,
                                                  save_every=self.save_every, eps=self.eps, seed=self.seed, traj_offset=traj_offset,
                                                  return_dlogp=bool(self.return_dlogp), rtol=self.rtol, atol=self.atol)
        self.n_fevals = res[-1]
        dlogp = C.like(res[1][:, :, None], x0s) if self.return_dlogp else None
        return C.like(res[0][:, :, None], x0s), dlogp
