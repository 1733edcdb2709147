"""Shared implementation of the ambient / latent cPaiNN shells and MoleculeIntegrator mirrors.

The reference batch is a PyG ``Batch``; here any object with the attributes the reference reads is accepted
(/root/reference/mdqm9/thermo/ambient/models/ode_wrapper.py:111-112, graph.py:27, embedding.py:78):
    x0 / x [N,3] f32, atoms | atom_number [N] i64, T0,T1 | T [N], edge_index [2,E] i64 (edge_index[0] = source),
    edge_type [E] i64, batch [N] i64.
Every batch is homogeneous (one species per run); ``split_batch`` verifies that and extracts the
per-molecule template the C ABI takes.
"""
from __future__ import annotations

import numpy as np

from .. import engine as _engine
from .. import synthetic as _syn
from .. import weights as _W
from . import _common as C

DEFAULT_TEMPS = [300, 400, 500, 600, 700, 800, 900, 1000]


def split_batch(batch, atom_key: str):
    """-> (B, A, edge_src[E_m], edge_dst[E_m], edge_type[E_m], atom_ids[A]); raises ValueError for heterogeneous batches."""
    bidx = C.to_numpy(batch.batch, np.int64)
    N = bidx.size
    if N == 0:
        raise ValueError("empty batch")
    B = int(bidx.max()) + 1
    if N % B or not np.array_equal(bidx, np.repeat(np.arange(B), N // B)):
        raise ValueError("batch.batch must be molecule-major with equally sized molecules (one species per run)")
    A = N // B
    ei = C.to_numpy(batch.edge_index, np.int64)
    et = C.to_numpy(batch.edge_type, np.int64)
    atoms = C.to_numpy(getattr(batch, atom_key), np.int64)
    if ei.ndim != 2 or ei.shape[0] != 2 or ei.shape[1] != et.size:
        raise ValueError("edge_index must be [2, E] and edge_type [E]")
    E = ei.shape[1]
    if E % B:
        raise ValueError("edge count is not a multiple of the number of molecules")
    Em = E // B
    off = (np.arange(B, dtype=np.int64) * A)[:, None]
    src = ei[0].reshape(B, Em) - off
    dst = ei[1].reshape(B, Em) - off
    ty = et.reshape(B, Em)
    at = atoms.reshape(B, A)
    if (src != src[0]).any() or (dst != dst[0]).any() or (ty != ty[0]).any() or (at != at[0]).any():
        raise ValueError("molecules of the batch differ in graph or atom ids; the sampler handles one species per batch")
    if Em and (src[0].min() < 0 or src[0].max() >= A or dst[0].min() < 0 or dst[0].max() >= A):
        raise ValueError("edges cross molecule boundaries")
    return B, A, src[0].astype(np.int32), dst[0].astype(np.int32), ty[0].astype(np.int32), at[0].astype(np.int32)


class PaiNNShell:
    """Weights-only stand-in for the reference ``cPaiNN`` modules (subclasses fix the variant)."""
    VARIANT = _W.AMBIENT
    ATOM_KEY = "atoms"
    COND_KEYS = ("T0", "T1")

    def _init(self, n_features, score_layers, n_types, temp_length, time_length, temperatures):
        self.n_features, self.score_layers, self.n_types = int(n_features), int(score_layers), int(n_types)
        self.temp_length, self.time_length = float(temp_length), float(time_length)
        self.temperatures = list(temperatures)
        self._spec = _W.painn_param_spec(self.VARIANT, self.n_features, self.score_layers, self.n_types)
        self._sd = _syn.make_state_dict(self._spec, seed=0)       # placeholder init; real weights come from load_state_dict
        self._engines, self._device = {}, 0
        self.precision = "f32"                  # 'f16x2': split-fp16 matrix path; 'f16': fp16 storage mode; set before first use
        self.training = False

    # -- torch.nn.Module surface used by the sampling drivers (sample_ambient.py:71-72,125-131)
    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, state_dict, strict=True):
        flat = _W.flatten_state_dict(state_dict, self._spec, strict=strict)
        self._sd = _W.unflatten(flat, self._spec)
        self._engines = {}
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, device=None, *a, **k):
        idx = getattr(device, "index", None)
        if isinstance(device, int):
            idx = device
        elif isinstance(device, str) and ":" in device:
            idx = int(device.split(":")[1])
        if idx is not None and idx != self._device:
            self._device, self._engines = idx, {}
        return self

    def parameters(self):
        return iter(self._sd.values())

    def engine_for(self, A, src, dst, ety, atom_ids) -> _engine.PainnEngine:
        key = (A, src.tobytes(), dst.tobytes(), ety.tobytes(), atom_ids.tobytes(), self.precision)
        if key not in self._engines:
            flat = _W.flatten_state_dict(self._sd, self._spec)
            self._engines[key] = _engine.PainnEngine(self.VARIANT, self.n_features, self.score_layers, A, src, dst, ety, atom_ids, flat,
                                                     n_types=self.n_types, temp_length=self.temp_length, time_length=self.time_length,
                                                     temperatures=self.temperatures, device=self._device, precision=self.precision)
        return self._engines[key]

    def cond_of(self, batch, B, A, on_gpu=False):
        """[B, A, n_cond] float32 conditioning; a CUDA tensor when `on_gpu` (the batch lives on the GPU), else numpy."""
        if not self.COND_KEYS:
            return None
        if on_gpu:
            import torch
            cols = [getattr(batch, k).detach().to(torch.float32).reshape(B, A) for k in self.COND_KEYS]
            return torch.stack(cols, dim=-1).contiguous()
        cols = [C.to_numpy(getattr(batch, k)).astype(np.float32).reshape(B, A) for k in self.COND_KEYS]   # latent T is int64 (mdqm9_latent.py:184)
        return np.ascontiguousarray(np.stack(cols, axis=-1))

    def forward(self, batch):
        """Evaluates the drift at batch.x, time batch.t (one value per call) and writes batch.output [N,3] like the reference."""
        B, A, src, dst, ety, ids = split_batch(batch, self.ATOM_KEY)
        t = C.to_numpy(batch.t, np.float64).ravel()
        if np.ptp(t) != 0.0:
            raise NotImplementedError("per-molecule times are a training-only input; the sampling path evaluates one t per call")
        x = C.as_f32(batch.x, (B, A, 3))                  # a CUDA batch is evaluated in place, no host round trip
        out = self.engine_for(A, src, dst, ety, ids).drift(x, float(t[0]), self.cond_of(batch, B, A, C.is_cuda(x)))
        batch.output = C.like(out.reshape(B * A, 3), batch.x)
        return batch

    __call__ = forward


class ODEWrapperBase:
    """Mirror of the reference ``ODEWrapper`` (mdqm9/thermo/{ambient,latent}/models/ode_wrapper.py): the right-hand side the
    integrator sees.  forward(t, states, batch[, n_steps]) -> b, or (b, -div * DIV_SCALE) with return_dlogp (reverse_ode:
    (-b, +div * DIV_SCALE)); compute_divergence(b, batch) -> div * DIV_SCALE evaluated at batch.x, batch.t."""
    DIV_SCALE = 1.0

    def __init__(self, b, return_dlogp=False, reverse_ode=False):
        self.b, self.return_dlogp, self.reverse_ode = b, return_dlogp, reverse_ode

    def _eval(self, batch, x, t, with_div):
        B, A, src, dst, ety, ids = split_batch(batch, self.b.ATOM_KEY)
        xs = C.as_f32(x, (B, A, 3))
        eng = self.b.engine_for(A, src, dst, ety, ids)
        cond = self.b.cond_of(batch, B, A, C.is_cuda(xs))
        if with_div:
            out, div = eng.drift_div(xs, float(t), cond)
            return out.reshape(B * A, 3), div
        return eng.drift(xs, float(t), cond).reshape(B * A, 3), None

    def forward(self, integration_time, states, batch, n_steps=None):
        if n_steps is not None:
            n_steps.append(n_steps[-1] + 1)                       # ambient wrapper's evaluation counter (ode_wrapper.py:44)
        t = float(C.to_numpy(integration_time).reshape(-1)[0])
        if self.return_dlogp:
            x, _ = states
            b, div = self._eval(batch, x, t, True)
            b, d = C.like(b, x), C.like(div * self.DIV_SCALE if C.is_torch(div) else div * np.float32(self.DIV_SCALE), x)
            return (b, -d) if not self.reverse_ode else (-b, d)
        b, _ = self._eval(batch, states, t, False)
        return C.like(b, states)

    __call__ = forward

    @classmethod
    def compute_divergence(cls, b, batch):
        t = float(C.to_numpy(batch.t).reshape(-1)[0])
        _, div = cls(b)._eval(batch, batch.x, t, True)
        return C.like(div * cls.DIV_SCALE if C.is_torch(div) else div * np.float32(cls.DIV_SCALE), batch.x)

    @staticmethod
    def reset_batch(batch, x, integration_time):
        batch.x = x.clone() if hasattr(x, "clone") else np.array(x, copy=True)
        ids = getattr(batch, "atoms", None)
        ids = getattr(batch, "atom_number") if ids is None else ids
        batch.t = integration_time * (ids * 0 + 1)                # integration_time * ones_like(atoms)  (ode_wrapper.py:112)
        return batch


class MoleculeIntegratorBase:
    """rollout(batch) with the reference's constructor.  method: 'dopri5' (the reference default; adaptive with rtol / atol, the
    grid linspace(start, end, n_step) selects the output times) or a scheme on that grid (see _common.check_method).

    return_dlogp=True integrates the second state of the reference ODEWrapper with the same scheme: d(dlogp)/dt = -DIV_SCALE *
    div b (exact divergence, 3A forward-mode passes per molecule on the GPU), returned * SCALE_DLOGP as [n_saved, B].  With
    reverse_ode the pair is (-b, +DIV_SCALE * div) on linspace(end, start) (ode_wrapper.py:49, integrators.py:40-43)."""
    SCALE_DLOGP = 1.0      # integrators.py:68 (ambient: 1e2)
    DIV_SCALE = 1.0        # ode_wrapper.py:91 (ambient: 1e-2)

    def __init__(self, b, method: str = "dopri5", n_step: int = 100, atol: float = 1e-4, rtol: float = 1e-4, start: float = 0.0,
                 end: float = 1.0, return_dlogp: bool = False, reverse_ode: bool = False, *, eps: float = 0.0, seed: int = 0,
                 save_every: int = 1, com_free_noise: bool = False):
        self.method = C.check_method(method)
        if return_dlogp and self.method == "em" and eps > 0:
            raise ValueError("return_dlogp=True needs a deterministic scheme ('euler' or 'heun')")
        self.b = b
        self.start, self.end, self.rtol, self.atol = start, end, rtol, atol
        self.n_step, self.return_dlogp, self.reverse_ode = n_step, return_dlogp, reverse_ode
        self.eps, self.seed, self.save_every, self.com_free_noise = eps, seed, save_every, com_free_noise

    def _rollout(self, batch, traj_offset=0):
        B, A, src, dst, ety, ids = split_batch(batch, self.b.ATOM_KEY)
        x0 = C.as_f32(batch.x0, (B, A, 3))               # CUDA batches stay in HBM: data_ptr() in, CUDA tensors out
        gpu = C.is_cuda(x0)
        # without dlogp the reference always integrates on linspace(start, end) (integrators.py:54-55), reverse_ode or not
        grid = _engine.time_grid(self.start, self.end, self.n_step)
        eng = self.b.engine_for(A, src, dst, ety, ids)
        if self.return_dlogp:
            if self.reverse_ode:
                grid = _engine.time_grid(self.end, self.start, self.n_step)
            path, dl, nfe = eng.rollout_dlogp(x0, self.b.cond_of(batch, B, A, gpu), grid, scheme="euler" if self.method == "em" else self.method,
                                              save_every=self.save_every, div_scale=self.DIV_SCALE, out_scale=self.SCALE_DLOGP,
                                              reverse_ode=self.reverse_ode, rtol=self.rtol, atol=self.atol)
            return C.like(path.reshape(path.shape[0], B * A, 3), batch.x0), C.like(dl, batch.x0), nfe
        path, nfe = eng.rollout(x0, self.b.cond_of(batch, B, A, gpu), grid, scheme=self.method, save_every=self.save_every, eps=self.eps,
                                seed=self.seed, traj_offset=traj_offset, com_free_noise=self.com_free_noise, rtol=self.rtol, atol=self.atol)
        xts = C.like(path.reshape(path.shape[0], B * A, 3), batch.x0)
        dlogp = batch.x0.new_zeros(B) if gpu else C.like(np.zeros(B, np.float32) * self.SCALE_DLOGP, batch.x0)      # reference: zeros(batch_size) (* 1e2 in ambient)
        return xts, dlogp, nfe
