"""Shared host-side helpers of the thermo/ mirror: tensor <-> numpy plumbing and solver-name handling."""
from __future__ import annotations

import numpy as np

FIXED_STEP = ("euler", "heun", "em")
# torchdiffeq method names the reference passes ('dopri5' in every shipped config).  The adaptive solvers live
# in a third-party package that is not part of the reference checkout; they are not reproduced (parity unpinned there).
ADAPTIVE = ("dopri5", "dopri8", "bosh3", "fehlberg2", "adaptive_heun", "rk4", "midpoint", "explicit_adams", "implicit_adams",
            "fixed_adams", "scipy_solver")


def check_method(method: str) -> str:
    if method in FIXED_STEP:
        return method
    if method in ADAPTIVE:
        raise NotImplementedError(
            f"method={method!r} is a torchdiffeq solver (third-party, absent from the reference checkout); this build provides the "
            f"fixed-step schemes {FIXED_STEP} on the same torch.linspace(start, end, n_step) grid.  'euler' equals torchdiffeq's "
            "method='euler' on that grid; use method='heun' for second order.")
    raise ValueError(f"unknown method {method!r}; expected one of {FIXED_STEP}")


def is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "detach")


def to_numpy(x, dtype=None) -> np.ndarray:
    if is_torch(x):
        x = x.detach().cpu().numpy()
    a = np.asarray(x)
    return a if dtype is None else a.astype(dtype, copy=False)


def like(result: np.ndarray, template):
    """Return `result` in the container type of `template` (torch tensor on the template's device, or numpy)."""
    if is_torch(template):
        import torch
        return torch.from_numpy(np.ascontiguousarray(result)).to(template.device)
    return result

