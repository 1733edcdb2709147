"""End-to-end physical check of the adw path: a drift TRAINED with the reference's objective
(tests/golden/train_adw.py -> adw_trained_h64.npz) carries Boltzmann samples of U = 4 (x^2 - 1)^2 + x / 2 from beta0 = 1.0 to
beta1 = 1.25, and the dlogp state makes importance weights  log w = -beta1 U(x1) + beta0 U(x0) - dlogp  whose self-normalised
averages reproduce the target ensemble whatever the quality of the training -- a test of the divergence, of its sign and of the
1e-2 / 1e2 scaling conventions that no parity fixture can give.  Tolerances are Monte-Carlo: 5 standard errors."""
import numpy as np
import pytest

from conftest import load_golden, pkg
from oracle import oracle

BETA0, BETA1 = 1.0, 1.25


def setup(n):
    ti = pkg()
    g = load_golden("adw_trained_h64")
    H, nl = int(g["hidden"]), int(g["num_layers"])
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    flat = ti.weights.flatten_state_dict(sd, ti.weights.adw_param_spec(H, nl), dtype=np.float64)
    x0 = ti.synthetic.adw_boltzmann(n, BETA0, seed=11)
    return ti, H, nl, sd, flat, x0


def check_reweighting(ti, x0, x1, dlogp):
    syn = ti.synthetic
    target, base = syn.adw_expectations(BETA1), syn.adw_expectations(BETA0)
    n = len(x0)
    logw = -BETA1 * syn.adw_potential(x1) + BETA0 * syn.adw_potential(x0) - dlogp
    w = np.exp(logw - logw.max())
    w /= w.sum()
    ess = 1.0 / (w ** 2).sum() / n
    assert ess > 0.85, ess                                              # measured 0.94; the wrong sign of dlogp gives 0.12
    for name, f in (("mean", x1), ("second", x1 ** 2), ("left", (x1 < 0).astype(np.float64))):
        est = (w * f).sum()
        se = np.sqrt((w ** 2 * (f - est) ** 2).sum())                   # delta-method standard error of the ratio estimator
        assert abs(est - target[name]) < 5 * se + 1e-4, (name, est, target[name], se)
    # free-energy profile  F(x) = -log p(x) / beta1  on a grid of bins (the adw end-state observable of the reference's
    # free_energy.py): reweighted histogram against the analytic profile, up to the additive constant, within 5 standard errors
    edges = np.linspace(-1.6, 1.6, 33)
    idx = np.digitize(x1, edges) - 1
    ok = (idx >= 0) & (idx < 32)
    pw = np.bincount(idx[ok], weights=w[ok], minlength=32)
    pw2 = np.bincount(idx[ok], weights=(w[ok] ** 2), minlength=32)
    centres = 0.5 * (edges[1:] + edges[:-1])
    fine = np.linspace(-1.6, 1.6, 32 * 64 + 1)
    pt = np.exp(-BETA1 * syn.adw_potential(fine))
    pt = 0.5 * (pt[1:] + pt[:-1])
    pt = pt.reshape(32, 64).sum(axis=1)
    keep = pw > 50.0 / n                                               # bins with enough effective counts
    f_est, f_true = -np.log(pw[keep]) / BETA1, -np.log(pt[keep]) / BETA1
    se = np.sqrt(pw2[keep]) / pw[keep] / BETA1
    shift = np.average(f_est - f_true, weights=1.0 / se ** 2)
    assert keep.sum() >= 20 and centres[keep].min() < -1.0 and centres[keep].max() > 1.0
    assert (np.abs(f_est - f_true - shift) < 5 * se + 2e-3).all(), np.abs(f_est - f_true - shift) / se
    # the transport itself moves the ensemble most of the way from the base to the target
    raw = x1.mean()
    assert abs(raw - target["mean"]) < 0.35 * abs(base["mean"] - target["mean"]), (raw, base["mean"], target["mean"])


def test_trained_drift_reweights_to_the_target_ensemble_cpu_oracle():
    ti, H, nl, sd, flat, x0 = setup(40000)
    orc = oracle.AdwOracle(H, nl, flat)
    grid = np.linspace(0, 1, 51).astype(np.float32)
    path, dl, _ = orc.rollout(x0, np.full(len(x0), BETA0), np.full(len(x0), BETA1), grid, scheme="heun", save_every=0, return_dlogp=True)
    check_reweighting(ti, x0, path[0], dl[0])


@pytest.mark.gpu
@pytest.mark.parametrize("method,precision", [("dopri5", "f32"), ("dopri5", "f16x2"), ("heun", "f32")])
def test_trained_drift_reweights_to_the_target_ensemble_gpu(method, precision):
    """The reference-facing call: StandardIntegrator(b, method='dopri5', rtol = atol = 1e-4, n_step, return_dlogp=True)."""
    torch = pytest.importorskip("torch")
    ti, H, nl, sd, flat, x0 = setup(400000)
    net = ti.thermo.adw.FCNetMultiBeta(1, 1, H, nl)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.precision = precision
    integ = ti.thermo.adw.StandardIntegrator(b=net, method=method, n_step=41, rtol=1e-4, atol=1e-4, return_dlogp=True, save_every=0)
    x0t = torch.from_numpy(x0.astype(np.float32))[:, None]
    sample, dlogp = integ.rollout(x0t, beta0s=torch.full_like(x0t, BETA0), beta1s=torch.full_like(x0t, BETA1))
    x1, dl = sample.numpy()[-1, :, 0].astype(np.float64), dlogp.numpy()[-1, :, 0].astype(np.float64)
    check_reweighting(ti, x0.astype(np.float32).astype(np.float64), x1, dl)
    # and the GPU trajectories agree with the CPU oracle's on a subset (dopri5: to its tolerance; heun: to round-off)
    if method == "heun":
        orc = oracle.AdwOracle(H, nl, flat)
        sub = slice(0, 2000)
        ref, rdl, _ = orc.rollout(x0.astype(np.float32)[sub], np.full(2000, BETA0), np.full(2000, BETA1), ti.engine.time_grid(0, 1, 41),
                                  scheme="heun", save_every=0, return_dlogp=True)
        assert np.abs(x1[sub] - ref[0]).max() < 2e-5 and np.abs(dl[sub] - rdl[0]).max() < 2e-4
