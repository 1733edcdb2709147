"""GPU parity of the forward-mode drift derivative, the exact divergence and the dlogp rollout.

Checked against (a) fixtures produced by the reference's own ODEWrapper(return_dlogp=True) autograd
(tests/golden/div_*.npz, make_golden.py) and (b) the CPU oracle's forward-mode twin, stage by stage.
Tolerances: a tangent is a linear image of the seed direction, so rel-L2 against the oracle uses the drift bar (1e-5);
the divergence is a signed sum of 3A Jacobian entries and is compared absolutely, |err| < 2e-5 (|div| + 1), the bar the
oracle itself is held to against the reference (tests/test_oracle_golden.py).
"""
import numpy as np
import pytest

from conftest import golden_weights, load_golden, pkg, rel_l2
from oracle import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5
DIV_ATOL = 2e-5
DIV_CASES = ["div_ambient_small", "div_ambient_sparse", "div_ambient_f128", "div_latent_multi", "div_latent_single"]


def make_pair(g, precision="f32"):
    ti = pkg()
    args = (int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"], g["atom_ids"], golden_weights(g))
    kw = dict(temp_length=float(g["temp_length"]), temperatures=g["temperatures"])
    return ti.engine.PainnEngine(*args, precision=precision, **kw), oracle.PainnOracle(*args, **kw)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", DIV_CASES)
def test_jvp_stage_taps_vs_oracle(name, precision):
    g = load_golden(name)
    eng, orc = make_pair(g, precision)
    B, L, t = int(g["B"]), int(g["L"]), float(g["t"])
    xdot = np.random.RandomState(5).standard_normal(g["x"].shape).astype(np.float32)
    b, tan = eng.jvp(g["x"], xdot, t, g["cond"])
    rb, rtan = orc.jvp(g["x"], xdot, t, g["cond"])
    assert rel_l2(b, rb) < TOL
    assert rel_l2(tan, rtan) < TOL, rel_l2(tan, rtan)
    stages = [s for l in range(L) for s in ((1 + 2 * l, f"msg{l}"), (2 + 2 * l, f"upd{l}"))]
    try:
        for stage, tag in stages:
            eng.debug_tap(stage)
            eng.jvp(g["x"], xdot, t, g["cond"])
            _, _, taps = orc.jvp(g["x"], xdot, t, g["cond"], tap_stage=stage)
            assert rel_l2(eng.debug_read("ts", B), taps["s"]) < TOL, (tag, "ts")
            assert rel_l2(eng.debug_read("tv", B).transpose(0, 1, 3, 2), taps["v"]) < TOL, (tag, "tv")
            if tag.startswith("msg") and int(tag[3:]) < L - 1:
                assert rel_l2(eng.debug_read("te", B), taps["e"]) < TOL, (tag, "te")
    finally:
        eng.debug_tap(-1)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_divergence_headline_shape_vs_reference_autograd(precision):
    """F = 128, L = 5, A = 18: divergence and two-state Euler / Heun step against the reference's autograd fixture."""
    g = load_golden("div_ambient_full")
    eng, _ = make_pair(g, precision)
    scale = float(g["div_scale"])
    ref_div = -g["negdiv_scaled"].astype(np.float64) / scale
    b, div = eng.drift_div(g["x"], float(g["t"]), g["cond"])
    assert rel_l2(b, g["drift"]) < TOL
    assert (np.abs(div - ref_div) < DIV_ATOL * (np.abs(ref_div) + 1.0)).all(), (div, ref_div)
    for scheme in ("euler", "heun"):
        path, dl, _ = eng.rollout_dlogp(g["x"], g["cond"], g["grid"], scheme=scheme, div_scale=scale)
        ref, ref_dl = g[f"traj_{scheme}"], g[f"dlogp_{scheme}"]
        assert rel_l2(path - path[0], ref - ref[0]) < 2e-5
        assert (np.abs(dl - ref_dl) < DIV_ATOL * scale * (np.abs(ref_dl) / scale + 1.0)).all(), (scheme, dl, ref_dl)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", DIV_CASES)
def test_divergence_vs_reference_autograd(name, precision):
    g = load_golden(name)
    eng, orc = make_pair(g, precision)
    scale = float(g["div_scale"])
    ref_div = -g["negdiv_scaled"].astype(np.float64) / scale
    b, div = eng.drift_div(g["x"], float(g["t"]), g["cond"])
    assert rel_l2(b, g["drift"]) < TOL
    assert (np.abs(div - ref_div) < DIV_ATOL * (np.abs(ref_div) + 1.0)).all(), (div, ref_div)
    _, odiv = orc.drift_div(g["x"], float(g["t"]), g["cond"], precision=64)
    assert (np.abs(div - odiv) < DIV_ATOL * (np.abs(odiv) + 1.0)).all()
    # the diagonal from unit seeds equals the JVP along each unit direction (same kernels, D = 1 path)
    B, A = int(g["B"]), int(g["A"])
    acc = np.zeros(B)
    for k in range(3 * A):
        xdot = np.zeros((B, A, 3), np.float32)
        xdot.reshape(B, -1)[:, k] = 1.0
        acc += eng.jvp(g["x"], xdot, float(g["t"]), g["cond"])[1].reshape(B, -1)[:, k]
    assert np.allclose(acc, div, rtol=0, atol=2e-6 * (np.abs(div).max() + 1.0))


@pytest.mark.parametrize("name", DIV_CASES)
@pytest.mark.parametrize("scheme", ["euler", "heun"])
def test_dlogp_rollout_vs_reference(name, scheme):
    g = load_golden(name)
    eng, _ = make_pair(g)
    scale = float(g["div_scale"])
    for rev in ([False, True] if "grid_rev" in g else [False]):
        tag = scheme + ("_rev" if rev else "")
        grid = g["grid_rev" if rev else "grid"]
        path, dl, nfe = eng.rollout_dlogp(g["x"], g["cond"], grid, scheme=scheme, div_scale=scale, out_scale=1.0, reverse_ode=rev)
        ref, ref_dl = g[f"traj_{tag}"], g[f"dlogp_{tag}"]
        assert path.shape == ref.shape and dl.shape == ref_dl.shape
        assert nfe == (len(grid) - 1) * (2 if scheme == "heun" else 1)
        assert rel_l2(path - path[0], ref - ref[0]) < 2e-5
        assert (np.abs(dl - ref_dl) < DIV_ATOL * scale * (np.abs(ref_dl) / scale + 1.0)).all(), (tag, dl, ref_dl)
        last, dl_last, _ = eng.rollout_dlogp(g["x"], g["cond"], grid, scheme=scheme, save_every=0, div_scale=scale, reverse_ode=rev)
        np.testing.assert_array_equal(last[0], path[-1])
        np.testing.assert_array_equal(dl_last[0], dl[-1])
    with pytest.raises(pkg()._lib.TiError):
        eng.rollout_dlogp(g["x"], g["cond"], g["grid"], scheme="em")


def test_divergence_full_size_chunked_and_ragged(monkeypatch):
    """BASELINE-size network (F=128, L=5, A=18): chunked tangent passes (tiny HBM budget -> several chunks, ragged last one)
    agree with the single-pass result bit for bit per molecule, and with the oracle on a sample."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 23
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    b, div = eng.drift_div(x, 0.5, cond)
    assert rel_l2(b, eng.drift(x, 0.5, cond)) == 0.0
    monkeypatch.setenv("TI_JVP_WS_GB", "0.12")          # ~14.5 MB of tangent state per molecule -> 8 molecules per pass
    b2, div2 = eng.drift_div(x, 0.5, cond)
    np.testing.assert_array_equal(b2, b)
    assert np.allclose(div2, div, rtol=0, atol=1e-5)    # group composition changes the order of the per-atom partial sums
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    idx = [0, 7, 8, B - 1]
    _, odiv = orc.drift_div(x[idx], 0.5, cond[idx], precision=64)
    assert (np.abs(div[idx] - odiv) < DIV_ATOL * (np.abs(odiv) + 1.0)).all(), (div[idx], odiv)


@pytest.mark.parametrize("name", ["div_ambient_small", "div_latent_multi"])
def test_molecule_integrator_return_dlogp(name):
    """The reference-facing API: MoleculeIntegrator(return_dlogp=True).rollout(batch) on a torch batch."""
    torch = pytest.importorskip("torch")
    from test_gpu_api import golden_batch, state_dict_of
    ti = pkg()
    g = load_golden(name)
    ambient = int(g["variant"]) == 0
    mod = ti.thermo.ambient if ambient else ti.thermo.latent
    kw = dict(n_features=int(g["F"]), score_layers=int(g["L"]), temp_length=int(g["temp_length"]))
    if not ambient:
        kw["temperatures"] = [int(x) for x in g["temperatures"]]
    b = mod.cPaiNN(**kw)
    b.load_state_dict(state_dict_of(g))
    batch = golden_batch(g, "atoms" if ambient else "atom_number")
    n_step = len(g["grid"])
    for rev in ([False, True] if "grid_rev" in g else [False]):
        integ = mod.MoleculeIntegrator(b=b, method="heun", n_step=n_step, atol=1e-5, rtol=1e-5, return_dlogp=True, reverse_ode=rev,
                                       start=0.0, end=1.0)
        res = integ.rollout(batch)
        xts, dlogp = res[0], res[1]
        tag = "heun_rev" if rev else "heun"
        ref = g[f"traj_{tag}"].reshape(n_step, -1, 3)
        out_scale = 1e2 if ambient else 1.0                             # ambient/integrators.py:68
        assert isinstance(dlogp, torch.Tensor) and tuple(dlogp.shape) == (n_step, int(g["B"]))
        assert rel_l2(xts.numpy() - ref[0], ref - ref[0]) < 2e-5
        ref_dl = g[f"dlogp_{tag}"] * out_scale
        assert np.allclose(dlogp.numpy(), ref_dl, rtol=0, atol=DIV_ATOL * (np.abs(ref_dl).max() + 1.0))


@pytest.mark.parametrize("F,L,A,B,variant,precision", [(256, 2, 5, 3, 0, "f32"), (256, 2, 5, 3, 0, "f16x2"), (64, 3, 25, 2, 1, "f32"),
                                                       (32, 2, 2, 7, 2, "f32"), (64, 2, 1, 4, 0, "f32")])
def test_divergence_other_widths_and_sizes_vs_oracle(F, L, A, B, variant, precision):
    """F = 256 (one workgroup per CU path), the largest molecule (25 atoms, 600 edges), a diatomic, and a single atom with no
    edges at all (the drift and its divergence vanish: v stays 0)."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
    x = syn.molecule_coords(B, A, seed=B)
    cond = [syn.ambient_cond(B, A), syn.latent_cond(B, A, 500.0), None][variant]
    kw = dict(temp_length=100.0 if variant == 0 else 75.0)
    eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, precision=precision, **kw)
    orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, **kw)
    b, div = eng.drift_div(x, 0.37, cond)
    ob, odiv = orc.drift_div(x, 0.37, cond, precision=64)
    assert np.isfinite(div).all()
    if A == 1:
        assert np.abs(b).max() == 0.0 and np.abs(div).max() == 0.0 and np.abs(odiv).max() == 0.0
        return
    assert rel_l2(b, ob) < TOL
    assert (np.abs(div - odiv) < DIV_ATOL * (np.abs(odiv) + 1.0)).all(), (div, odiv)


@pytest.mark.parametrize("template", ["throughput", "latency"])
def test_divergence_race_screen_full_occupancy(template, monkeypatch):
    """256 molecules x 54 directions fill every CU with two workgroups of each tangent kernel.  During development the
    split-fp16 build of the tangent readout kernel returned wrong sums for about one 16-node tile in 2 000 in exactly this
    regime (5-10 molecules per evaluation off by 1e-3..1e-1) while every small-batch parity test passed; the f32 and split
    builds are independent instruction streams, so per-molecule agreement of repeated evaluations of both is the screen."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    monkeypatch.setenv("TI_TEMPLATE", template)
    F, L, A, B = 128, 2, 18, 256
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=F + A), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=B), syn.ambient_cond(B, A)
    outs = []
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        outs += [eng.drift_div(x, 0.5, cond)[1].astype(np.float64) for _ in range(4)]
        eng.close()
    ref = outs[0]
    for o in outs[1:]:
        bad = np.abs(o - ref) > 5e-5 * (np.abs(ref) + 1.0)
        assert not bad.any(), f"{int(bad.sum())} molecules disagree, worst {np.abs(o - ref).max():.2e}"


@pytest.mark.parametrize("name", ["div_ambient_small", "div_latent_multi"])
def test_ode_wrapper_mirror(name):
    """ODEWrapper(b, return_dlogp=True)(t, (x, dlogp), batch[, n_steps]) -> (b, -div * scale) like the reference module."""
    torch = pytest.importorskip("torch")
    from test_gpu_api import golden_batch, state_dict_of
    ti = pkg()
    g = load_golden(name)
    ambient = int(g["variant"]) == 0
    mod = ti.thermo.ambient if ambient else ti.thermo.latent
    kw = dict(n_features=int(g["F"]), score_layers=int(g["L"]), temp_length=int(g["temp_length"]))
    if not ambient:
        kw["temperatures"] = [int(x) for x in g["temperatures"]]
    b = mod.cPaiNN(**kw)
    b.load_state_dict(state_dict_of(g))
    batch = golden_batch(g, "atoms" if ambient else "atom_number")
    ode = mod.ODEWrapper(b, return_dlogp=True)
    n_steps = [0]
    args = (torch.tensor(float(g["t"])), (batch.x0.clone(), torch.zeros(int(g["B"]))), batch) + ((n_steps,) if ambient else ())
    drift, negdiv = ode(*args)
    assert rel_l2(drift.numpy().reshape(g["drift"].shape), g["drift"]) < TOL
    scale = float(g["div_scale"])
    assert np.allclose(negdiv.numpy(), g["negdiv_scaled"], rtol=0, atol=DIV_ATOL * scale * (np.abs(g["negdiv_scaled"]).max() / scale + 1.0))
    if ambient:
        assert n_steps == [0, 1]
    rev = mod.ODEWrapper(b, return_dlogp=True, reverse_ode=True)(*args[:3])
    assert np.allclose(rev[0].numpy(), -drift.numpy()) and np.allclose(rev[1].numpy(), -negdiv.numpy())
    plain = mod.ODEWrapper(b)(torch.tensor(float(g["t"])), batch.x0.clone(), batch)
    assert rel_l2(plain.numpy().reshape(g["drift"].shape), g["drift"]) < TOL
