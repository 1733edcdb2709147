"""Sampling drivers with the reference's loop structure and output files.

  sample_ambient <- /root/reference/mdqm9/sample_ambient.py:18-119   samples_/dlogps_/latent_noises_/latent_dlogps_{name}.npy
  sample_latent  <- /root/reference/mdqm9/sample_latent.py:19-96     samples_/dlogps_{name}_forward.npy (dlogps with return_dlogp)
  sample_adw     <- /root/reference/adw/sample.py:14-81              initial_samples/samples/dlogps _epoch_{k}.npy under beta_{b0}_to_{b1}/
  load_config    <- /root/reference/mdqm9/thermo/utils.py:31-47      JSON file -> argparse.Namespace (keys become --options)

Differences, on purpose: `method` comes from ``config.method`` (default 'dopri5' like the reference's hard-coded solver; any
name of thermo._common.SUPPORTED works); files are written once per call instead of re-saving the growing concatenation after every batch
(O(n_batches^2) I/O in the reference); datasets are the numpy ones of ``data.py``.
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np

from .thermo import adw as _adw
from .thermo import ambient as _amb
from .thermo import latent as _lat
from .thermo import _common as C


def load_config(path: str, filename: str, argv=()) -> argparse.Namespace:
    settings = json.load(open(os.path.join(path, filename)))
    parser = argparse.ArgumentParser()
    for key, value in settings.items():
        parser.add_argument(f"--{key}", type=type(value), default=value)
    return parser.parse_args(list(argv))


def _regroup(sample, batch_idx):
    """[n_step, N, 3] -> [B, n_step, A, 3] exactly like ``np.array([sample[:, batch_idx == i] ...])`` (sample_ambient.py:93)."""
    sample, batch_idx = C.to_numpy(sample), C.to_numpy(batch_idx)
    return np.array([sample[:, batch_idx == i] for i in range(int(batch_idx.max()) + 1)])


def sample_ambient(config, b, dataset):
    os.makedirs(config.data_save_path, exist_ok=True)
    integrator = _amb.MoleculeIntegrator(b=b, method=getattr(config, "method", "dopri5"), rtol=config.rtol, atol=config.atol,
                                         n_step=config.n_steps, return_dlogp=bool(config.return_dlogp), reverse_ode=False,
                                         save_every=getattr(config, "save_every", 1))
    latent_noises, latent_dlogps, samples, dlogps, n_fevals = [], [], [], [], 0
    b.eval()
    for batch in dataset.batches(config.batch_size, shuffle=True, seed=config.seed):
        bidx = batch.batch
        latent_noises.append(np.array([batch.latent_z[bidx == i] for i in range(int(bidx.max()) + 1)]))
        latent_dlogps.append(batch.latent_dlogp)
        sample, dlogp, n_fevals, _ = integrator.rollout(batch)
        samples.append(_regroup(sample, bidx))
        if config.return_dlogp:
            dlogps.append(C.to_numpy(dlogp)[-1, :])
    name = config.data_save_name
    np.save(os.path.join(config.data_save_path, f"latent_noises_{name}.npy"), np.concatenate(latent_noises, axis=0))
    np.save(os.path.join(config.data_save_path, f"latent_dlogps_{name}.npy"), np.concatenate(latent_dlogps, axis=0))
    np.save(os.path.join(config.data_save_path, f"samples_{name}.npy"), np.concatenate(samples, axis=0))
    if config.return_dlogp:
        np.save(os.path.join(config.data_save_path, f"dlogps_{name}.npy"), np.concatenate(dlogps, axis=0))
    return np.concatenate(samples, axis=0), n_fevals


def sample_latent(config, b, dataset):
    os.makedirs(config.data_save_path, exist_ok=True)
    integrator = _lat.MoleculeIntegrator(b=b, method=getattr(config, "method", "dopri5"), rtol=config.rtol, atol=config.atol,
                                         n_step=config.n_steps, return_dlogp=bool(config.return_dlogp), reverse_ode=False,
                                         save_every=getattr(config, "save_every", 1))
    samples, dlogps = [], []
    b.eval()
    for batch in dataset.batches(config.batch_size, seed=config.seed, drop_last=True):
        sample, dlogp, bidx = integrator.rollout(batch)
        samples.append(_regroup(sample, bidx))
        if config.return_dlogp:
            dlogps.append(C.to_numpy(dlogp)[-1, :])                     # sample_latent.py:76-77
    out = np.concatenate(samples, axis=0)
    np.save(os.path.join(config.data_save_path, f"samples_{config.data_save_name}_forward.npy"), out)
    if config.return_dlogp:
        np.save(os.path.join(config.data_save_path, f"dlogps_{config.data_save_name}_forward.npy"), np.concatenate(dlogps, axis=0))
    return out


def sample_adw(config, b, x0s_batches):
    """`x0s_batches`: iterable of (x0s [B,1], beta0s [B,1]) like the reference's test loader (adw/sample.py:41-43)."""
    assert len(config.beta0s) == len(config.beta1s) == 1            # adw/sample.py:24
    integrator = _adw.StandardIntegrator(b=b, method=getattr(config, "method", None) or config.solver_type, rtol=config.rtol,
                                         atol=config.atol, n_step=config.n_step, return_dlogp=bool(config.return_dlogp))
    initial, samples, dlogps = [], [], []
    b.eval()
    for x0s, beta0s in x0s_batches:
        beta1s = np.ones_like(C.to_numpy(beta0s)) * config.beta1s[0]
        sample, dlogp = integrator.rollout(x0s, beta0s=beta0s, beta1s=beta1s)
        initial.append(C.to_numpy(x0s))
        samples.append(C.to_numpy(sample))
        if config.return_dlogp:
            dlogps.append(C.to_numpy(dlogp))
    out_dir = os.path.join(config.data_save_path, config.model_save_name, f"beta_{config.beta0s[0]}_to_{config.beta1s[0]}")
    os.makedirs(out_dir, exist_ok=True)
    initial = np.array(initial)[:, :, 0].flatten()
    np.save(os.path.join(out_dir, f"initial_samples_epoch_{config.sampling_epoch}.npy"), initial)

    def by_step(chunks):                                             # [n_batches, n_step, B, 1] -> [n_step, n_batches*B]
        arr = np.array(chunks)
        return np.array([arr[:, i, :, 0].flatten() for i in range(arr.shape[1])])

    samples = by_step(samples)
    np.save(os.path.join(out_dir, f"samples_epoch_{config.sampling_epoch}.npy"), samples)
    if config.return_dlogp:
        np.save(os.path.join(out_dir, f"dlogps_epoch_{config.sampling_epoch}.npy"), by_step(dlogps))
    return initial, samples
