"""GPU: the Runge-Kutta drivers of libti_hip.so (dopri5 / midpoint / rk4 restating torchdiffeq 0.2.5, csrc/ti_api.hip rollout_rk)
against the numpy restatement of the same algorithm over the CPU oracle drift (oracle/ode.py).  torchdiffeq is absent:
parity with the library itself is UNPINNED.  Tolerances: fixed-grid schemes follow the drift bar accumulated
over the steps (2e-5 rel-L2 on the displacement); dopri5 may take a different accept/reject path when an error ratio sits
within fp32 noise of 1, so it is held to 20x its own tolerance in max-norm and to the evaluation-count structure."""
import numpy as np
import pytest

from conftest import golden_weights, load_golden, pkg, rel_l2
from oracle import ode, oracle

pytestmark = pytest.mark.gpu


def painn_pair(g, precision="f32"):
    ti = pkg()
    args = (int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"], g["atom_ids"], golden_weights(g))
    kw = dict(temp_length=float(g["temp_length"]), temperatures=g["temperatures"])
    return ti.engine.PainnEngine(*args, precision=precision, **kw), oracle.PainnOracle(*args, **kw)


@pytest.mark.parametrize("name", ["ambient_small", "latent_multi", "ambient_a9"])
@pytest.mark.parametrize("scheme", ["midpoint", "rk4"])
def test_fixed_grid_schemes_vs_restatement(name, scheme):
    g = load_golden(name)
    eng, orc = painn_pair(g)
    grid = g["traj_grid"]
    path, nfe = eng.rollout(g["x"], g["cond"], grid, scheme=scheme)
    sol, nfe_ref = ode.odeint(lambda t, y: [orc.drift(y[0], t, g["cond"])], [g["x"]], grid, scheme)
    assert nfe == nfe_ref == (len(grid) - 1) * (2 if scheme == "midpoint" else 4)
    assert path.shape == sol[0].shape
    assert rel_l2(path - path[0], sol[0] - sol[0][0]) < 2e-5
    # higher order than Euler on the same grid: closer to a fine Heun solution
    fine, _ = orc.rollout(g["x"], g["cond"], np.linspace(0, 1, 40 * (len(grid) - 1) + 1).astype(np.float32), scheme="heun", save_every=0)
    euler, _ = eng.rollout(g["x"], g["cond"], grid, scheme="euler")
    assert np.abs(path[-1] - fine[0]).max() < 0.5 * np.abs(euler[-1] - fine[0]).max()


@pytest.mark.parametrize("name", ["ambient_small", "latent_multi"])
@pytest.mark.parametrize("tol", [1e-4, 1e-6])
def test_dopri5_vs_restatement(name, tol):
    g = load_golden(name)
    eng, orc = painn_pair(g)
    grid = np.linspace(0.0, 1.0, 7).astype(np.float32)
    path, nfe = eng.rollout(g["x"], g["cond"], grid, scheme="dopri5", rtol=tol, atol=tol)
    sol, nfe_ref = ode.odeint(lambda t, y: [orc.drift(y[0], t, g["cond"])], [g["x"]], grid, "dopri5", tol, tol)
    assert path.shape == sol[0].shape and (nfe - 2) % 6 == 0
    # the two drift implementations differ by fp32 round-off (~1e-6 relative per evaluation) on top of the tolerance
    assert np.abs(path - sol[0]).max() < 20 * tol + 2e-5 * np.abs(g["x"]).max()
    assert abs(nfe - nfe_ref) <= 12                                        # at most two differing accept/reject decisions
    np.testing.assert_array_equal(path[0], g["x"])
    again, nfe2 = eng.rollout(g["x"], g["cond"], grid, scheme="dopri5", rtol=tol, atol=tol)
    np.testing.assert_array_equal(again, path)                              # deterministic (fixed-order reductions)
    assert nfe2 == nfe
    last, _ = eng.rollout(g["x"], g["cond"], grid, scheme="dopri5", rtol=tol, atol=tol, save_every=0)
    np.testing.assert_array_equal(last[0], path[-1])
    fine, _ = orc.rollout(g["x"], g["cond"], np.linspace(0, 1, 601).astype(np.float32), scheme="heun", save_every=100)
    assert np.abs(path - fine).max() < (5e-3 if tol > 1e-5 else 2e-4)       # measured 2.1e-3 / 4e-5 on the CPU restatement


def test_dopri5_with_dlogp_and_reverse():
    """Two-state run (x, dlogp): mixed norm over the pair, second state -div_scale * div; reverse_ode on the descending grid."""
    g = load_golden("div_latent_multi")
    ti = pkg()
    args = (int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"], g["atom_ids"], golden_weights(g))
    kw = dict(temp_length=float(g["temp_length"]), temperatures=g["temperatures"])
    eng, orc = ti.engine.PainnEngine(*args, **kw), oracle.PainnOracle(*args, **kw)
    tol = 1e-5

    def rhs(sign):
        def f(t, y):
            b, div = orc.drift_div(y[0], t, g["cond"])
            return [sign * b, (-sign * div).astype(np.float32)]
        return f

    for rev, grid in ((False, np.linspace(0, 1, 4)), (True, np.linspace(1, 0, 4))):
        grid = grid.astype(np.float32)
        path, dl, nfe = eng.rollout_dlogp(g["x"], g["cond"], grid, scheme="dopri5", rtol=tol, atol=tol, reverse_ode=rev)
        sol, nfe_ref = ode.odeint(rhs(-1.0 if rev else 1.0), [g["x"], np.zeros(int(g["B"]), np.float32)], grid, "dopri5", tol, tol)
        assert np.abs(path - sol[0]).max() < 20 * tol and np.abs(dl - sol[1]).max() < 20 * tol * (np.abs(sol[1]).max() + 1)
        assert abs(nfe - nfe_ref) <= 12


def test_adw_dopri5_and_default_integrator():
    torch = pytest.importorskip("torch")
    ti = pkg()
    g = load_golden("adw_ctor_h64")
    H, nl = int(g["hidden"]), int(g["num_layers"])
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    flat = ti.weights.flatten_state_dict(sd, ti.weights.adw_param_spec(H, nl), dtype=np.float64)
    eng, orc = ti.engine.AdwEngine(H, nl, flat), oracle.AdwOracle(H, nl, flat)
    x0, b0, b1 = g["x"].astype(np.float32), g["beta0"].astype(np.float32), g["beta1"].astype(np.float32)
    grid = np.linspace(0.0, 1.0, 11).astype(np.float32)
    tol = 1e-5
    path, dl, nfe = eng.rollout(x0, b0, b1, grid, scheme="dopri5", rtol=tol, atol=tol, return_dlogp=True)

    def f(t, y):
        b, div = orc.drift_div(y[0].astype(np.float64), t, b0.astype(np.float64), b1.astype(np.float64))
        return [b.astype(np.float32), (-div * 1e-2).astype(np.float32)]

    with pytest.raises(ti._lib.TiError):                       # dlogp needs a deterministic flow
        eng.rollout(x0, b0, b1, grid, scheme="em", eps=0.1, return_dlogp=True)
    sol, nfe_ref = ode.odeint(f, [x0, np.zeros_like(x0)], grid, "dopri5", tol, tol)
    assert np.abs(path - sol[0]).max() < 20 * tol and np.abs(dl - sol[1] * 1e2).max() < 20 * tol * 1e2
    assert abs(nfe - nfe_ref) <= 12
    # the mirror class with the reference's default arguments (method='dopri5', rtol = atol = 1e-4)
    net = ti.thermo.adw.FCNetMultiBeta(1, 1, H, nl)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    integ = ti.thermo.adw.StandardIntegrator(b=net, n_step=11, return_dlogp=True)
    sample, dlogp = integ.rollout(torch.from_numpy(x0)[:, None], beta0s=torch.from_numpy(b0)[:, None], beta1s=torch.from_numpy(b1)[:, None])
    assert tuple(sample.shape) == (11, len(x0), 1) and tuple(dlogp.shape) == (11, len(x0), 1)
    ref = g["traj_heun"]                                     # reference-module Heun trajectory on the same 11-point grid
    assert np.abs(sample.numpy()[:, :, 0] - ref).max() < 5e-3
