"""Weight loading: reference ``state_dict`` -> the canonical flat fp32/fp64 buffer of ``include/ti_hip.h``.

This is the one place PyTorch tensors are accepted (``.detach().cpu().numpy()``); everything below the
C ABI sees a flat array.  The canonical order is documented in ``include/ti_hip.h`` (section "weight
layout") and is shared by the HIP library and the CPU oracle.

Reference key layout (measured from ``state_dict()`` of the reference constructors):
  ambient cPaiNN  (/root/reference/mdqm9/thermo/ambient/models/cpainn.py:67-90)
      net.2 edge-type embedding, net.3 atom embedding, net.7 combine MLP, net.8 PaiNNBase
  latent cPaiNN   (/root/reference/mdqm9/thermo/latent/models/cpainn.py:43-72)
      multi-T : net.2, net.3, net.6 combine MLP, net.7 PaiNNBase
      single-T: net.2, net.3, net.5 combine MLP, net.6 PaiNNBase
  FCNetMultiBeta  (/root/reference/adw/thermo/models/simple.py:21-36)
      net.{0,2,...}, beta_embed.{0,2,4}
``device_tracker`` scalars (/root/reference/mdqm9/thermo/ambient/models/device.py:18-26) are accepted
and ignored.
"""
from __future__ import annotations

import numpy as np

AMBIENT, LATENT_MULTI, LATENT_SINGLE = 0, 1, 2
VARIANT_NAMES = {AMBIENT: "ambient", LATENT_MULTI: "latent_multi", LATENT_SINGLE: "latent_single"}
N_EMBED = {AMBIENT: 4, LATENT_MULTI: 3, LATENT_SINGLE: 2}          # F-wide invariant embeddings fed to the combine MLP
N_COND = {AMBIENT: 2, LATENT_MULTI: 1, LATENT_SINGLE: 0}           # per-node conditioning scalars (T0,T1 | T | -)
_COMBINE_IDX = {AMBIENT: 7, LATENT_MULTI: 6, LATENT_SINGLE: 5}
N_EDGE_TYPES = 4


def _mlp_spec(prefix: str, f_in: int, f_h: int, f_out: int):
    """torch.nn.Sequential(Linear, LayerNorm, SiLU, Linear, LayerNorm, SiLU, Linear)
    (/root/reference/mdqm9/thermo/ambient/models/embedding.py:27-35)."""
    return [
        (f"{prefix}.0.weight", (f_h, f_in)), (f"{prefix}.0.bias", (f_h,)),
        (f"{prefix}.1.weight", (f_h,)), (f"{prefix}.1.bias", (f_h,)),
        (f"{prefix}.3.weight", (f_h, f_h)), (f"{prefix}.3.bias", (f_h,)),
        (f"{prefix}.4.weight", (f_h,)), (f"{prefix}.4.bias", (f_h,)),
        (f"{prefix}.6.weight", (f_out, f_h)), (f"{prefix}.6.bias", (f_out,)),
    ]


def painn_param_spec(variant: int, F: int, L: int, n_types: int = 25):
    """Ordered list of (reference state_dict key, shape) = the canonical flat layout."""
    ci = _COMBINE_IDX[variant]
    base = f"net.{ci + 1}.layers"
    spec = [("net.2.embedding.weight", (N_EDGE_TYPES, F)), ("net.3.embedding.weight", (n_types, F))]
    spec += _mlp_spec(f"net.{ci}.mlp.mlp", N_EMBED[variant] * F, F, F)
    for l in range(L):
        spec += _mlp_spec(f"{base}.{2 * l}.phi.mlp", 2 * F, F, 5 * F)
        spec += _mlp_spec(f"{base}.{2 * l}.w.mlp", F, F, 5 * F)
        spec += [(f"{base}.{2 * l + 1}.u.linear.weight", (F, F)), (f"{base}.{2 * l + 1}.v.linear.weight", (F, F))]
        spec += _mlp_spec(f"{base}.{2 * l + 1}.mlp.mlp", 2 * F, F, 3 * F)
    spec += _mlp_spec(f"{base}.{2 * L}.mlp.mlp", F, F, 2)
    spec += [(f"{base}.{2 * L}.V.linear.weight", (1, F))]
    return spec


def adw_param_spec(hidden: int, num_layers: int, in_size: int = 1, out_size: int = 1):
    """FCNetMultiBeta (/root/reference/adw/thermo/models/simple.py:21-36): beta_embed first, then net."""
    spec = [("beta_embed.0.weight", (hidden, 3)), ("beta_embed.0.bias", (hidden,)),
            ("beta_embed.2.weight", (hidden, hidden)), ("beta_embed.2.bias", (hidden,)),
            ("beta_embed.4.weight", (1, hidden)), ("beta_embed.4.bias", (1,))]
    sizes = [in_size + 2] + [hidden] * num_layers + [out_size]
    for i in range(len(sizes) - 1):
        spec += [(f"net.{2 * i}.weight", (sizes[i + 1], sizes[i])), (f"net.{2 * i}.bias", (sizes[i + 1],))]
    return spec


def n_params(spec) -> int:
    return int(sum(int(np.prod(s)) for _, s in spec))


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):            # torch.Tensor without importing torch here
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def flatten_state_dict(state_dict, spec, dtype=np.float32, strict: bool = True) -> np.ndarray:
    """Concatenate ``state_dict`` tensors in canonical order.  Mirrors ``load_state_dict(strict=True)``:
    missing keys / shape mismatches / unexpected keys raise ``RuntimeError`` (device_tracker keys are skipped)."""
    wanted = dict(spec)
    missing = [k for k in wanted if k not in state_dict]
    unexpected = [k for k in state_dict if k not in wanted and not k.endswith("device_tracker")]
    if missing or (strict and unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing}, unexpected keys {unexpected}")
    parts = []
    for k, shape in spec:
        a = _to_numpy(state_dict[k])
        if tuple(a.shape) != tuple(shape):
            raise RuntimeError(f"size mismatch for {k}: got {tuple(a.shape)}, expected {tuple(shape)}")
        parts.append(np.ascontiguousarray(a, dtype=dtype).ravel())
    return np.concatenate(parts)


def unflatten(flat: np.ndarray, spec) -> dict:
    out, o = {}, 0
    for k, shape in spec:
        n = int(np.prod(shape))
        out[k] = flat[o:o + n].reshape(shape)
        o += n
    assert o == flat.size, (o, flat.size)
    return out
