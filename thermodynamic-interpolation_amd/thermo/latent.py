"""Drop-in mirror of the reference latent sampling API (thermo/latent) on top of libti_hip.so.

  cPaiNN             <- /root/reference/mdqm9/thermo/latent/models/cpainn.py:10-108
  MoleculeIntegrator <- /root/reference/mdqm9/thermo/latent/integrators.py:8-89
"""
from __future__ import annotations

from .. import weights as _W
from ._molecule import DEFAULT_TEMPS, MoleculeIntegratorBase, ODEWrapperBase, PaiNNShell


class cPaiNN(PaiNNShell):
    ATOM_KEY = "atom_number"

    def __init__(self, n_features: int = 32, score_layers: int = 5, n_types=25, time_length=10, temp_length=10, temperatures=DEFAULT_TEMPS):
        # one temperature feature T, or none when the model knows a single temperature (cpainn.py:43-72)
        multi = len(temperatures) > 1
        self.VARIANT = _W.LATENT_MULTI if multi else _W.LATENT_SINGLE
        self.COND_KEYS = ("T",) if multi else ()
        self._init(n_features, score_layers, n_types, temp_length, time_length, temperatures)


class ODEWrapper(ODEWrapperBase):
    """thermo/latent/models/ode_wrapper.py:6-113"""
    DIV_SCALE = 1.0


class MoleculeIntegrator(MoleculeIntegratorBase):
    """rollout(batch) -> (xts [n_saved, N, 3], dlogp, batch.batch)   (integrators.py:89; no 1e2 scaling in latent)"""
    SCALE_DLOGP = 1.0

    def rollout(self, batch, traj_offset: int = 0):
        xts, dlogp, self.n_fevals = self._rollout(batch, traj_offset)
        return xts, dlogp, batch.batch
