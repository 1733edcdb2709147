"""The numpy restatement of the torchdiffeq 0.2.5 solvers (oracle/ode.py) on problems with closed-form solutions, and on the
oracle drift networks against a fine-step reference.  CPU only.  torchdiffeq itself is absent: parity with the library is
UNPINNED; what is pinned here is that the restated algorithm solves ODEs to its tolerance and has the library's structure
(6 evaluations per attempted step, FSAL, 2 evaluations for the initial step)."""
import numpy as np
import pytest

from conftest import golden_weights, load_golden
from oracle import ode, oracle


def test_dopri5_closed_form_and_structure():
    t = np.linspace(0.0, 2.0, 9)
    for tol in (1e-3, 1e-5, 1e-7):
        sol, nfe = ode.odeint(lambda tt, y: [y[0] * np.float32(np.cos(tt))], [np.float32([1.0, 0.5])], t, "dopri5", tol, tol)
        exact = np.outer(np.exp(np.sin(t)), [1.0, 0.5])
        assert np.abs(sol[0] - exact).max() < 20 * tol + 3e-6
        assert (nfe - 2) % 6 == 0                                # f0 + initial-step probe + 6 per attempted step
    # two-entry state with very different scales: the mixed norm makes the tighter-behaved entry control the step
    sol, _ = ode.odeint(lambda tt, y: [-y[0], np.float32(100.0) * np.ones_like(y[1])], [np.ones(3, np.float32), np.zeros(2, np.float32)],
                        t, "dopri5", 1e-5, 1e-5)
    assert np.abs(sol[0][:, 0] - np.exp(-t)).max() < 1e-4 and np.abs(sol[1][:, 0] - 100.0 * t).max() < 1e-2
    # decreasing grid = integrating back: forward then backward returns to the start
    fwd, _ = ode.odeint(lambda tt, y: [y[0] * np.float32(np.cos(tt))], [np.float32([1.0])], t, "dopri5", 1e-6, 1e-6)
    back, _ = ode.odeint(lambda tt, y: [y[0] * np.float32(np.cos(tt))], [fwd[0][-1]], t[::-1].copy(), "dopri5", 1e-6, 1e-6)
    assert abs(back[0][-1, 0] - 1.0) < 1e-4


@pytest.mark.parametrize("method,order", [("midpoint", 2), ("rk4", 4)])
def test_fixed_grid_orders(method, order):
    errs = []
    for n in (11, 21):
        t = np.linspace(0.0, 1.0, n)
        sol, nfe = ode.odeint(lambda tt, y: [-2.0 * y[0]], [np.ones(1, np.float32)], t, method)
        errs.append(abs(float(sol[0][-1, 0]) - np.exp(-2.0)))
        assert nfe == (n - 1) * (2 if method == "midpoint" else 4)
    assert errs[0] / errs[1] > 0.6 * 2 ** order


def test_dopri5_on_the_drift_network_matches_fine_heun():
    g = load_golden("ambient_small")
    o = oracle.PainnOracle(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                           g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"])
    f = lambda tt, y: [o.drift(y[0], tt, g["cond"])]
    t = np.linspace(0.0, 1.0, 6).astype(np.float32)
    fine, _ = o.rollout(g["x"], g["cond"], np.linspace(0.0, 1.0, 401).astype(np.float32), scheme="heun", save_every=80)
    errs = []
    for tol in (1e-4, 1e-5, 1e-6):              # measured: 2.1e-3, 5.6e-4, 4.1e-5 with 14, 20, 38 evaluations
        sol, nfe = ode.odeint(f, [g["x"]], t, "dopri5", tol, tol)
        assert sol[0].shape == fine.shape and nfe < 100
        errs.append(float(np.abs(sol[0] - fine).max()))
    assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-4
