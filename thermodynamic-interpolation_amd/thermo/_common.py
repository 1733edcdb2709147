"""Shared host-side helpers of the thermo/ mirror: tensor <-> numpy plumbing and solver-name handling."""
from __future__ import annotations

import numpy as np

# Solver names: torchdiffeq's 'dopri5' (the reference default), 'euler', 'midpoint', 'rk4' restate torchdiffeq
# 0.2.5 (ti_env.yml:14; third-party, absent from the reference checkout -- parity unpinned against the library itself);
# 'heun' and 'em' are build-defined (include/ti_hip.h).
SUPPORTED = ("dopri5", "euler", "midpoint", "rk4", "heun", "em")
NOT_BUILT = ("dopri8", "bosh3", "fehlberg2", "adaptive_heun", "heun2", "heun3", "explicit_adams", "implicit_adams", "fixed_adams",
             "scipy_solver")


def check_method(method: str) -> str:
    if method in SUPPORTED:
        return method
    if method in NOT_BUILT:
        raise NotImplementedError(
            f"method={method!r} is a torchdiffeq solver that is not built here (torchdiffeq is third-party and absent from the "
            f"reference checkout); available: {SUPPORTED}.")
    raise ValueError(f"unknown method {method!r}; expected one of {SUPPORTED}")


def is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "detach")


def is_cuda(x) -> bool:
    return is_torch(x) and bool(x.is_cuda)


def as_f32(x, shape):
    """`x` as a float32, C-contiguous [shape] buffer WHERE IT LIVES: a CUDA tensor stays on the GPU (the engine then uses it
    in place through data_ptr(), no host copy), anything else becomes a numpy array."""
    if is_cuda(x):
        import torch
        return x.detach().to(torch.float32).reshape(shape).contiguous()
    return np.ascontiguousarray(to_numpy(x, np.float32).reshape(shape))


def to_numpy(x, dtype=None) -> np.ndarray:
    if is_torch(x):
        x = x.detach().cpu().numpy()
    a = np.asarray(x)
    return a if dtype is None else a.astype(dtype, copy=False)


def like(result: np.ndarray, template):
    """Return `result` in the container type of `template` (torch tensor on the template's device, or numpy)."""
    if is_torch(result):                          # device-resident path: already a tensor where the inputs live
        return result
    if is_torch(template):
        import torch
        return torch.from_numpy(np.ascontiguousarray(result)).to(template.device)
    return result

