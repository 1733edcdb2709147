"""Host-side logic that needs no GPU: weight loading, time grid, batch splitting, API mirror argument handling."""
import types

import numpy as np
import pytest

from conftest import load_golden, pkg


def test_time_grid_equals_torch_linspace():
    torch = pytest.importorskip("torch")
    ti = pkg()
    for a, b, n in [(0.0, 1.0, 2), (0.0, 1.0, 11), (0.0, 1.0, 100), (0.0, 1.0, 401), (0.0, 1.0, 1001), (1.0, 0.0, 100), (0.2, 0.9, 7), (0.0, 1.0, 1)]:
        ref = torch.linspace(a, b, n).numpy()
        np.testing.assert_array_equal(ti.engine.time_grid(a, b, n), ref)
        np.testing.assert_allclose(ti.engine._time_grid_numpy(a, b, n), ref, rtol=0, atol=1.2e-7)      # torch-free fallback: 1 ulp


def test_state_dict_loader_is_strict_and_ignores_device_trackers():
    ti = pkg()
    g = load_golden("ambient_ctor")                        # state_dict of the reference constructor, all keys
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    assert any(k.endswith("device_tracker") for k in sd)
    spec = ti.weights.painn_param_spec(0, int(g["F"]), int(g["L"]), 25)
    flat = ti.weights.flatten_state_dict(sd, spec)
    assert flat.size == ti.weights.n_params(spec) and flat.dtype == np.float32
    assert set(sd) - {k for k in sd if k.endswith("device_tracker")} == {k for k, _ in spec}
    bad = dict(sd)
    bad.pop("net.3.embedding.weight")
    with pytest.raises(RuntimeError, match="missing keys"):
        ti.weights.flatten_state_dict(bad, spec)
    bad = dict(sd, extra=np.zeros(3))
    with pytest.raises(RuntimeError, match="unexpected keys"):
        ti.weights.flatten_state_dict(bad, spec)
    bad = dict(sd)
    bad["net.2.embedding.weight"] = np.zeros((5, int(g["F"])), np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        ti.weights.flatten_state_dict(bad, spec)
    # latent key layouts: combine MLP at net.6 (multi-T) / net.5 (single-T)
    gl = load_golden("latent_ctor")
    sdl = {k[4:]: v for k, v in gl.items() if k.startswith("sd::")}
    ti.weights.flatten_state_dict(sdl, ti.weights.painn_param_spec(1, int(gl["F"]), int(gl["L"]), 25))
    assert "net.5.mlp.mlp.0.weight" in dict(ti.weights.painn_param_spec(2, 32, 2))


def test_param_counts_match_survey():
    ti = pkg()
    W = ti.weights
    assert W.n_params(W.painn_param_spec(0, 128, 5)) == 2040845 - 11        # SURVEY §2 count includes the 11 device_tracker scalars (6 + L)
    assert W.n_params(W.adw_param_spec(256, 5)) == 331522


def make_batch(B, A, template, variant=0):
    ti = pkg()
    src, dst, et = template
    b = types.SimpleNamespace()
    b.x0 = ti.synthetic.molecule_coords(B, A, 0).reshape(B * A, 3)
    b.x = b.x0.copy()
    b.edge_index = ti.synthetic.batch_edge_index(src, dst, A, B)
    b.edge_type = np.tile(et.astype(np.int64), B)
    b.batch = np.repeat(np.arange(B), A)
    b.atoms = np.tile(np.arange(A), B)
    b.atom_number = b.atoms
    b.T0 = np.full(B * A, 1000.0, np.float32)
    b.T1 = np.full(B * A, 300.0, np.float32)
    b.T = np.full(B * A, 800, np.int64)
    return b


def test_split_batch_extracts_template_and_rejects_heterogeneous_batches():
    ti = pkg()
    mol = ti.thermo._molecule
    tmpl = ti.synthetic.sparse_template(7, seed=3)
    b = make_batch(4, 7, tmpl)
    B, A, src, dst, ety, ids = mol.split_batch(b, "atoms")
    assert (B, A) == (4, 7)
    np.testing.assert_array_equal(src, tmpl[0])
    np.testing.assert_array_equal(dst, tmpl[1])
    np.testing.assert_array_equal(ety, tmpl[2])
    np.testing.assert_array_equal(ids, np.arange(7))
    bad = make_batch(4, 7, tmpl)
    bad.edge_type = bad.edge_type.copy()
    bad.edge_type[-1] ^= 1
    with pytest.raises(ValueError, match="differ"):
        mol.split_batch(bad, "atoms")
    bad = make_batch(4, 7, tmpl)
    bad.batch = bad.batch[::-1].copy()
    with pytest.raises(ValueError, match="molecule-major"):
        mol.split_batch(bad, "atoms")
    bad = make_batch(2, 7, tmpl)
    bad.edge_index = bad.edge_index.copy()
    bad.edge_index[0, 0] = 9                               # an edge into the second molecule
    with pytest.raises(ValueError):
        mol.split_batch(bad, "atoms")


def test_api_mirror_signatures_and_solver_names():
    ti = pkg()
    amb, lat, adw = ti.thermo.ambient, ti.thermo.latent, ti.thermo.adw
    b = amb.cPaiNN(n_features=32, score_layers=2, temp_length=100)
    assert b.VARIANT == 0 and amb.cPaiNN(n_features=32).score_layers == 5
    assert lat.cPaiNN(n_features=32, score_layers=2, temperatures=[800]).VARIANT == 2
    assert lat.cPaiNN(n_features=32, score_layers=2).VARIANT == 1
    for cls in (amb.MoleculeIntegrator, lat.MoleculeIntegrator):
        integ = cls(b=b, method="heun", rtol=1e-5, atol=1e-5, n_step=100, return_dlogp=False, reverse_ode=False)
        assert integ.n_step == 100
        assert cls(b=b, n_step=10).method == "dopri5"                  # the reference default is built (restated torchdiffeq 0.2.5)
        with pytest.raises(NotImplementedError, match="torchdiffeq"):
            cls(b=b, method="dopri8", n_step=10)
        assert cls(b=b, method="euler", n_step=10, return_dlogp=True, reverse_ode=True).return_dlogp       # exact divergence: built
        with pytest.raises(ValueError, match="deterministic"):
            cls(b=b, method="em", n_step=10, return_dlogp=True, eps=0.1)
        with pytest.raises(ValueError):
            cls(b=b, method="rk45")
    net = adw.FCNetMultiBeta(1, 1, 64, 3)
    g = load_golden("adw_ctor_h64")
    net.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith("sd::")})
    assert net.state_dict()["net.0.weight"].shape == (64, 3)
    adw.StandardIntegrator(b=net, method="euler", rtol=1e-4, atol=1e-4, n_step=400, return_dlogp=False)
    assert adw.StandardIntegrator(b=net, n_step=400).method == "dopri5"
    with pytest.raises(NotImplementedError):
        adw.StandardIntegrator(b=net, method="bosh3", n_step=400)
    with pytest.raises(NotImplementedError):
        adw.FCNetMultiBeta(2, 2, 64, 3)


def test_synthetic_weights_are_reproducible_and_nontrivial():
    ti = pkg()
    a = ti.synthetic.painn_state_dict(0, 32, 2, 25, 7)
    b = ti.synthetic.painn_state_dict(0, 32, 2, 25, 7)
    c = ti.synthetic.painn_state_dict(0, 32, 2, 25, 8)
    assert all(np.array_equal(a[k], b[k]) for k in a) and any(not np.array_equal(a[k], c[k]) for k in a)
    g = a["net.7.mlp.mlp.1.weight"]
    assert abs(g.mean() - 1) < 0.1 and g.std() > 0.01       # LayerNorm gamma is perturbed away from 1


def test_as_ptr_refuses_what_the_kernels_would_misread():
    """The kernels read / write float32 C-contiguous memory of the documented shape: other dtypes are refused instead of being
    reinterpreted, and a numpy `out` that would need a copy is refused instead of filling a temporary."""
    ti = pkg()
    A = ti._lib.as_ptr
    x = np.zeros((4, 3, 3), np.float64)
    p, keep, dev, idx = A(x, shape=(4, 3, 3))                       # inputs are converted (a copy is fine for an input)
    assert keep.dtype == np.float32 and not dev and idx is None
    with pytest.raises(ValueError):
        A(np.zeros((4, 3, 2), np.float32), shape=(4, 3, 3), what="x")
    with pytest.raises(TypeError):
        A(np.zeros((4, 3, 3), np.float64), shape=(4, 3, 3), out=True)
    with pytest.raises(TypeError):
        A(np.zeros((4, 3, 6), np.float32)[:, :, ::2], shape=(4, 3, 3), out=True)          # not contiguous: would be copied
    ok = np.zeros((4, 3, 3), np.float32)
    assert A(ok, shape=(4, 3, 3), out=True)[1] is ok
    torch = pytest.importorskip("torch")
    with pytest.raises(TypeError):
        A(torch.zeros(4, 3, 3, dtype=torch.float64))
    with pytest.raises(TypeError):
        A(torch.zeros(4, 3, 3, dtype=torch.float16))
    with pytest.raises(ValueError):
        A(torch.zeros(4, 3, 6)[:, :, ::2])
    with pytest.raises(ValueError):
        A(torch.zeros(4, 3, 3), shape=(4, 3, 2))
    assert A(torch.zeros(4, 3, 3), shape=(4, 3, 3))[2] is False


def test_rollout_desc_carries_the_noise_counter_offset():
    ti = pkg()
    rd = ti.engine._rollout_desc("em", np.linspace(0, 1, 5), 0, 0, 0.1, 3, 100, False, step_offset=40)
    assert rd.step_offset == 40 and rd.n_step == 5 and rd.traj_offset == 100
