"""End-state observables of the mdqm9 path (BASELINE north star: "Boltzmann-weighted RMSD histogram ... agree within sampling
error").  No trained mdqm9 model or trajectory data exists offline, so the observable is computed on the SAME initial batch
with the same synthetic weights by the HIP path and by the CPU oracle: heavy-atom-free Kabsch RMSD of every end state to a
reference frame, histogrammed with the importance weights exp(-dlogp) the divergence provides.  Because the trajectories agree
to round-off the two histograms must agree far inside any sampling error; the test states that numerically."""
import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import oracle

pytestmark = pytest.mark.gpu


def kabsch_rmsd(x, ref):
    """x [B,A,3], ref [A,3]: minimal RMSD over rotations after centring."""
    x = x - x.mean(axis=1, keepdims=True)
    ref = ref - ref.mean(axis=0, keepdims=True)
    h = np.einsum("bai,aj->bij", x, ref)
    u, s, vt = np.linalg.svd(h)
    d = np.sign(np.linalg.det(u @ vt))
    s[:, -1] *= d
    e0 = (x ** 2).sum(axis=(1, 2)) + (ref ** 2).sum()
    return np.sqrt(np.maximum(e0 - 2.0 * s.sum(axis=1), 0.0) / x.shape[1])


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_weighted_rmsd_histogram_matches_cpu_path(precision):
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 32, 2, 9, 96
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=3), W.painn_param_spec(0, F, L, 25))
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=precision)
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x0, cond = syn.molecule_coords(B, A, seed=5), syn.ambient_cond(B, A)
    grid = ti.engine.time_grid(0.0, 1.0, 9)
    path, dl, _ = eng.rollout_dlogp(x0, cond, grid, scheme="heun", save_every=0, div_scale=1e-2, out_scale=1e2)
    opath, odl, _ = orc.rollout_dlogp(x0, cond, grid, scheme="heun", save_every=0, div_scale=1e-2)
    assert rel_l2(path[0] - x0, opath[0] - x0) < 2e-5
    assert np.abs(dl[0] - odl[0] * 1e2).max() < 1e-3 * (np.abs(odl[0] * 1e2).max() + 1.0)

    def observable(x1, dlogp):
        r = kabsch_rmsd(x1.astype(np.float64), x0[0].astype(np.float64))
        w = np.exp(-(dlogp - dlogp.min()))
        hist, _ = np.histogram(r, bins=12, range=(0.0, r.max() * 1.0001 + 1e-9), weights=w / w.sum())
        return r, hist

    r_g, h_g = observable(path[0], dl[0].astype(np.float64))
    r_c, h_c = observable(opath[0], odl[0].astype(np.float64) * 1e2)
    assert np.abs(r_g - r_c).max() < 1e-5
    assert np.abs(h_g - h_c).sum() < 5e-3            # a molecule within round-off of a bin edge may move its (small) weight


def test_headline_shape_observable_and_em_seeds():
    """The same observable at the bench shape (F = 128, L = 5, A = 18; reference observables: mdqm9/analysis/results_00031.py:140-149,
    weights utils/ess.py:8-35): 4 096 molecules on the GPU (two Euler steps with the exact divergence, 54 forward-mode directions per
    molecule), a 64-molecule subsample of the end states and an 8-molecule subsample of dlogp against the CPU oracle, the weighted
    RMSD histogram with the oracle's values substituted for the subsample; then two Euler-Maruyama runs with different seeds whose
    weighted histograms agree within their bootstrap error."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 4096
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16x2")
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x0, cond = syn.molecule_coords(B, A, seed=11), syn.ambient_cond(B, A)
    grid = ti.engine.time_grid(0.0, 1.0, 3)
    path, dl, nfe = eng.rollout_dlogp(x0, cond, grid, scheme="euler", save_every=0, div_scale=1e-2, out_scale=1e2)
    assert nfe == 2 and np.isfinite(path).all() and np.isfinite(dl).all()
    i64 = np.arange(0, B, B // 64)
    opath, _ = orc.rollout(x0[i64], cond[i64], grid, scheme="euler", save_every=0)
    assert rel_l2(path[0][i64] - x0[i64], opath[0] - x0[i64]) < 2e-5
    i8 = i64[::8]
    _, odl, _ = orc.rollout_dlogp(x0[i8], cond[i8], grid, scheme="euler", save_every=0, div_scale=1e-2)
    assert np.abs(dl[0][i8] - odl[0] * 1e2).max() < 1e-3 * (np.abs(odl[0] * 1e2).max() + 1.0)

    ref = x0[0].astype(np.float64)

    def hist(x1, logw, bins):
        r = kabsch_rmsd(x1.astype(np.float64), ref)
        w = np.exp(-(logw - logw.min()))
        return np.histogram(r, bins=bins, weights=w / w.sum())[0]

    r_all = kabsch_rmsd(path[0].astype(np.float64), ref)
    bins = np.linspace(0.0, r_all.max() * 1.0001 + 1e-9, 17)
    logw = dl[0].astype(np.float64)
    h_gpu = hist(path[0], logw, bins)
    x_mix, lw_mix = path[0].copy(), logw.copy()
    x_mix[i64] = opath[0]
    lw_mix[i8] = odl[0] * 1e2
    assert np.abs(h_gpu - hist(x_mix, lw_mix, bins)).sum() < 5e-3

    # Euler-Maruyama, two seeds: the stochastic sampler's weighted histogram is reproducible within its own sampling error
    grid8 = ti.engine.time_grid(0.0, 1.0, 9)
    ends = [eng.rollout(x0, cond, grid8, scheme="em", eps=0.02, seed=s, save_every=0)[0][0] for s in (1, 2)]
    assert np.abs(ends[0] - ends[1]).max() > 1e-4                      # the seeds really differ
    r = [kabsch_rmsd(e.astype(np.float64), ref) for e in ends]
    bins = np.linspace(0.0, max(r[0].max(), r[1].max()) * 1.0001 + 1e-9, 17)
    w = np.exp(-(logw - logw.min())); w /= w.sum()                     # importance weights of the initial configurations
    h = [np.histogram(ri, bins=bins, weights=w)[0] for ri in r]
    rs = np.random.RandomState(0)
    boot = []
    for _ in range(200):
        k = rs.randint(0, B, B)
        wk = w[k] / w[k].sum()
        boot.append(np.histogram(r[0][k], bins=bins, weights=wk)[0] - np.histogram(r[1][k], bins=bins, weights=wk)[0])
    sigma = np.std(boot, axis=0) + 1e-4
    assert (np.abs(h[0] - h[1]) < 5.0 * sigma + 1e-3).all(), (h[0] - h[1], sigma)
