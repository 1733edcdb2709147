"""Deterministic synthetic weights, graphs and inputs (no datasets / checkpoints exist offline).

Everything is a pure function of an integer seed through ``numpy.random.RandomState`` (legacy, stream-stable),
so the golden generator (tests/golden/make_golden.py, which loads these weights INTO the reference modules),
the CPU oracle, the HIP library and bench.py all see bit-identical weights and inputs without shipping them.
Shapes follow SURVEY.md §8(d) "Synthetic inputs".
"""
from __future__ import annotations

import numpy as np

from . import weights as W


def _fill(rs: np.random.RandomState, key: str, shape, fan_in_hint=None) -> np.ndarray:
    if "embedding.weight" in key:                       # torch.nn.Embedding ~ N(0,1)
        return rs.standard_normal(shape)
    if len(shape) == 2:                                  # Linear weight ~ U(+-1/sqrt(fan_in))
        b = 1.0 / np.sqrt(shape[1])
        return rs.uniform(-b, b, shape)
    # 1-D: LayerNorm gamma/beta or Linear bias.  In an MLP block '.1.'/.4.' are LayerNorms.
    tail = key.split(".")[-2:]
    if tail[0] in ("1", "4") and ".mlp." in key:
        if tail[1] == "weight":
            return 1.0 + 0.1 * rs.uniform(-1, 1, shape)  # gamma != 1 so a dropped affine shows up in parity
        return 0.1 * rs.uniform(-1, 1, shape)            # beta != 0
    b = 1.0 / np.sqrt(fan_in_hint or shape[0])
    return rs.uniform(-b, b, shape)


def make_state_dict(spec, seed: int, dtype=np.float32) -> dict:
    rs = np.random.RandomState(seed)
    sd, last_fan_in = {}, None
    for key, shape in spec:
        if len(shape) == 2:
            last_fan_in = shape[1]
        sd[key] = np.ascontiguousarray(_fill(rs, key, shape, last_fan_in), dtype=dtype)
    return sd


def painn_state_dict(variant: int, F: int, L: int, n_types: int = 25, seed: int = 0) -> dict:
    return make_state_dict(W.painn_param_spec(variant, F, L, n_types), seed)


def scale_state_dict(sd: dict, recipe) -> dict:
    """Multiply every tensor whose key ends with one of the recipe's suffixes by that factor: recipe = [(suffix, factor), ...].
    Used by the range fixtures (tests/golden/range_*.npz store the recipe, not the weights): e.g. scaling the edge embedding, the
    last Linear of the embed MLP and the message output layers drives the un-normalised streams s, e, |Vv| to 1e3..1e4 or down
    to 1e-6..1e-9, the magnitudes that matter for the split-fp16 operand path."""
    out = {}
    for k, v in sd.items():
        f = 1.0
        for suffix, factor in recipe:
            if k.endswith(suffix):
                f *= float(factor)
        out[k] = (v * np.asarray(f, v.dtype)).astype(v.dtype) if f != 1.0 else v
    return out


def adw_state_dict(hidden: int = 256, num_layers: int = 5, seed: int = 0, dtype=np.float64) -> dict:
    return make_state_dict(W.adw_param_spec(hidden, num_layers), seed, dtype)


# ----------------------------------------------------------------------------------------------- graphs
def fully_connected_template(A: int):
    """All ordered pairs i != j sorted by (src, dst) as PyG ``coalesce`` does
    (/root/reference/mdqm9/thermo/utils.py:69-81); edge_type 1 for |i-j| == 1 ("chain bonds"), else 0."""
    src, dst = np.meshgrid(np.arange(A), np.arange(A), indexing="ij")
    m = src != dst
    src, dst = src[m].astype(np.int32), dst[m].astype(np.int32)
    etype = (np.abs(src - dst) == 1).astype(np.int32)
    return src, dst, etype


def sparse_template(A: int, seed: int = 0, keep: float = 0.4):
    """Symmetric sparse graph (a finite ``cutoff``): chain bonds always kept (types 1..3), other pairs kept
    with probability ``keep`` (type 0).  Sorted by (src, dst).  Node A-1 keeps only its chain bond (degree 1)."""
    rs = np.random.RandomState(seed)
    adj = np.zeros((A, A), np.int32) - 1
    for i in range(A - 1):
        adj[i, i + 1] = adj[i + 1, i] = 1 + (i % 3)
    for i in range(A - 1):
        for j in range(i + 2, A - 1):
            if rs.uniform() < keep:
                adj[i, j] = adj[j, i] = 0
    src, dst = np.nonzero(adj >= 0)
    return src.astype(np.int32), dst.astype(np.int32), adj[src, dst].astype(np.int32)


def batch_edge_index(src, dst, A: int, B: int) -> np.ndarray:
    """[2, B*E_m] int64 global edge index, molecule-major (what a PyG DataLoader collation produces)."""
    off = (np.arange(B, dtype=np.int64) * A)[:, None]
    return np.stack([(src[None, :] + off).ravel(), (dst[None, :] + off).ravel()]).astype(np.int64)


# ----------------------------------------------------------------------------------------------- inputs
LADDER = (300.0, 400.0, 500.0, 600.0, 700.0, 800.0)


def molecule_coords(B: int, A: int, seed: int = 0, sigma: float = 0.3) -> np.ndarray:
    """x0 ~ N(0, sigma^2) per coordinate, centre of mass removed per molecule
    (mirrors /root/reference/mdqm9/data/mdqm9_ambient.py:161-162).  Returns [B, A, 3] float32."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, A, 3)) * sigma
    x -= x.mean(axis=1, keepdims=True)
    return x.astype(np.float32)


def ambient_cond(B: int, A: int, T0: float = 1000.0, ladder=LADDER) -> np.ndarray:
    """[B, A, 2] float32: T0 fixed, T1 round-robin over the ladder (trajectory i gets rung i mod len)."""
    c = np.empty((B, A, 2), np.float32)
    c[..., 0] = T0
    c[..., 1] = np.asarray(ladder, np.float32)[np.arange(B) % len(ladder)][:, None]
    return c


def latent_cond(B: int, A: int, T: float = 800.0) -> np.ndarray:
    return np.full((B, A, 1), T, np.float32)


def adw_x0(B: int, seed: int = 0) -> np.ndarray:
    return np.random.RandomState(seed).standard_normal(B).astype(np.float32)


def adw_potential(x):
    """The asymmetric double well of the adw experiments, U(x) = 4 (x^2 - 1)^2 + x / 2."""
    x = np.asarray(x, np.float64)
    return 4.0 * (x * x - 1.0) ** 2 + 0.5 * x


def adw_boltzmann(B: int, beta: float, seed: int = 0, lo: float = -3.0, hi: float = 3.0, n_grid: int = 60001) -> np.ndarray:
    """B independent samples of exp(-beta U) by inverse-CDF on a fine grid (stands in for the reference's samples.csv, which is
    not available offline)."""
    grid = np.linspace(lo, hi, n_grid)
    pdf = np.exp(-beta * (adw_potential(grid) - adw_potential(grid).min()))
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]))])
    cdf /= cdf[-1]
    return np.interp(np.random.RandomState(seed).random_sample(B), cdf, grid)


def adw_expectations(beta: float, lo: float = -3.0, hi: float = 3.0, n_grid: int = 60001) -> dict:
    """Quadrature values of <x>, <x^2>, P(x < 0) under exp(-beta U)."""
    grid = np.linspace(lo, hi, n_grid)
    w = np.exp(-beta * (adw_potential(grid) - adw_potential(grid).min()))
    w /= w.sum()
    return {"mean": float((w * grid).sum()), "second": float((w * grid * grid).sum()), "left": float(w[grid < 0].sum())}
