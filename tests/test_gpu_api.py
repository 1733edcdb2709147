"""The thermo/ mirror (reference class names and call signatures) on the GPU, against the reference golden vectors.
These read like the reference's own sampling drivers (mdqm9/sample_ambient.py:55-93, adw/sample.py:29-45)."""
import types

import numpy as np
import pytest

from conftest import load_golden, pkg, rel_l2

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def golden_batch(g, atom_key):
    ti = pkg()
    B, A = int(g["B"]), int(g["A"])
    b = types.SimpleNamespace()
    b.x0 = torch.from_numpy(g["x"].reshape(B * A, 3).copy())
    b.x = b.x0.clone()
    b.edge_index = torch.from_numpy(ti.synthetic.batch_edge_index(g["edge_src"], g["edge_dst"], A, B))
    b.edge_type = torch.from_numpy(np.tile(g["edge_type"].astype(np.int64), B))
    b.batch = torch.arange(B).repeat_interleave(A)
    setattr(b, atom_key, torch.from_numpy(np.tile(g["atom_ids"].astype(np.int64), B)))
    if int(g["variant"]) == 0:
        b.T0 = torch.from_numpy(g["cond"][..., 0].reshape(-1).copy())
        b.T1 = torch.from_numpy(g["cond"][..., 1].reshape(-1).copy())
    elif int(g["variant"]) == 1:
        b.T = torch.from_numpy(g["cond"][..., 0].reshape(-1).astype(np.int64))      # int64 like mdqm9_latent.py:184
    return b


def state_dict_of(g):
    ti = pkg()
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")}
    return sd or {k: torch.from_numpy(v) for k, v in ti.synthetic.painn_state_dict(int(g["variant"]), int(g["F"]), int(g["L"]), 25, int(g["seed"])).items()}


@pytest.mark.parametrize("name", ["ambient_ctor", "ambient_small", "ambient_sparse"])
def test_ambient_cpainn_and_integrator(name):
    ti = pkg()
    g = load_golden(name)
    b = ti.thermo.ambient.cPaiNN(n_features=int(g["F"]), score_layers=int(g["L"]), temp_length=int(g["temp_length"]))
    b.load_state_dict(state_dict_of(g))            # reference key layout, incl. ignored device_tracker entries
    b.eval()
    b.to(torch.device("cuda:0"))
    batch = golden_batch(g, "atoms")
    for i, t in enumerate(g["ts"]):
        batch.t = float(t) * torch.ones_like(batch.atoms)      # ODEWrapper.reset_batch (ode_wrapper.py:112)
        out = b(batch).output
        assert isinstance(out, torch.Tensor) and tuple(out.shape) == (int(g["B"]) * int(g["A"]), 3)
        assert rel_l2(out.numpy().reshape(g[f"drift_{i}"].shape), g[f"drift_{i}"]) < 1e-5
    if "traj_heun" in g:
        n_step = len(g["traj_grid"])
        integ = ti.thermo.ambient.MoleculeIntegrator(b=b, method="heun", rtol=1e-5, atol=1e-5, n_step=n_step, return_dlogp=False, reverse_ode=False)
        xts, dlogp, n_fevals, bidx = integ.rollout(batch)
        assert tuple(xts.shape) == (n_step, int(g["B"]) * int(g["A"]), 3) and n_fevals == 2 * (n_step - 1)
        assert tuple(dlogp.shape) == (int(g["B"]),) and float(dlogp.abs().max()) == 0.0 and bidx is batch.batch
        ref = g["traj_heun"].reshape(n_step, -1, 3)
        assert rel_l2(xts.numpy() - ref[0], ref - ref[0]) < 2e-5


@pytest.mark.parametrize("name", ["latent_ctor", "latent_multi", "latent_single"])
def test_latent_cpainn_and_integrator(name):
    ti = pkg()
    g = load_golden(name)
    b = ti.thermo.latent.cPaiNN(n_features=int(g["F"]), score_layers=int(g["L"]), temp_length=int(g["temp_length"]),
                                temperatures=[int(x) for x in g["temperatures"]])
    b.load_state_dict(state_dict_of(g))
    batch = golden_batch(g, "atom_number")
    batch.t = float(g["ts"][1]) * torch.ones_like(batch.atom_number)
    assert rel_l2(b(batch).output.numpy().reshape(g["drift_1"].shape), g["drift_1"]) < 1e-5
    if "traj_euler" in g:
        n_step = len(g["traj_grid"])
        integ = ti.thermo.latent.MoleculeIntegrator(b=b, method="euler", n_step=n_step, atol=1e-5, rtol=1e-5)
        xts, dlogp, bidx = integ.rollout(batch)                     # latent returns a 3-tuple (integrators.py:89)
        ref = g["traj_euler"].reshape(n_step, -1, 3)
        assert rel_l2(xts.numpy() - ref[0], ref - ref[0]) < 2e-5


def test_adw_fcnet_and_standard_integrator():
    ti = pkg()
    g = load_golden("adw_ctor_h64")
    net = ti.thermo.adw.FCNetMultiBeta(1, 1, int(g["hidden"]), int(g["num_layers"]))
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")})
    net.eval()
    x0s = torch.from_numpy(g["x"])[:, None]
    beta0s = torch.from_numpy(g["beta0"])[:, None]
    beta1s = torch.ones_like(beta0s) * 1.25                                  # adw/sample.py:43
    ts = torch.ones_like(x0s) * float(g["ts"][1])
    out = net(x0s, x0s, ts, beta0s, beta1s)
    assert tuple(out.shape) == (len(g["x"]), 1)
    assert rel_l2(out.numpy()[:, 0], g["drift_1"]) < 1e-5
    n_step = len(g["traj_grid"])
    integ = ti.thermo.adw.StandardIntegrator(b=net, method="euler", rtol=1e-4, atol=1e-4, n_step=n_step, return_dlogp=False)
    sample, dlogp = integ.rollout(x0s, beta0s=beta0s, beta1s=beta1s)
    assert tuple(sample.shape) == (n_step, len(g["x"]), 1) and dlogp is None
    assert rel_l2(sample.numpy()[:, :, 0], g["traj_euler"]) < 1e-5
