"""Drop-in mirror of the reference ambient sampling API (thermo/ambient) on top of libti_hip.so.

  cPaiNN             <- /root/reference/mdqm9/thermo/ambient/models/cpainn.py:10-115
  MoleculeIntegrator <- /root/reference/mdqm9/thermo/ambient/integrators.py:8-68
"""
from __future__ import annotations

from .. import weights as _W
from ._molecule import DEFAULT_TEMPS, MoleculeIntegratorBase, ODEWrapperBase, PaiNNShell


class cPaiNN(PaiNNShell):
    VARIANT, ATOM_KEY, COND_KEYS = _W.AMBIENT, "atoms", ("T0", "T1")

    def __init__(self, n_features: int = 32, embedding_layers: int = 2, score_layers: int = 5, n_types=25, temp_length=10, time_length=10,
                 temperatures=DEFAULT_TEMPS):
        self.embedding_layers = embedding_layers          # unused by the reference as well (dead constructor argument)
        self._init(n_features, score_layers, n_types, temp_length, time_length, temperatures)


class ODEWrapper(ODEWrapperBase):
    """thermo/ambient/models/ode_wrapper.py:6-113"""
    DIV_SCALE = 1e-2


class MoleculeIntegrator(MoleculeIntegratorBase):
    """rollout(batch) -> (xts [n_saved, N, 3], dlogp * 1e2, n_fevals, batch.batch)   (integrators.py:68)"""
    SCALE_DLOGP = 1e2
    DIV_SCALE = 1e-2

    def rollout(self, batch, traj_offset: int = 0):
        xts, dlogp, nfe = self._rollout(batch, traj_offset)
        return xts, dlogp, nfe, batch.batch
