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
