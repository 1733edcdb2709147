"""GPU parity of the PAIR-MAJOR message kernel (csrc/painn_pair_kernel.hpp; include/ti_hip.h TI_TEMPLATE_PAIR): the filter branch
w(enc(|r_ij|)) of SE3Message (reference cpainn.py:283-289) evaluated once per atom pair, the phi branch for both directions in lock
step.  Same bars as tests/test_gpu_parity.py: drift rel-L2 < 1e-5 against the reference fixtures, stage taps against the reference
intermediates, bit-identical re-runs.  Needs a real MI355X: `pytest -m gpu`.
"""
import numpy as np
import pytest

from conftest import golden_weights, load_golden, pkg, rel_l2
from oracle import oracle
from test_gpu_parity import DRIFT_TOL, PAINN_CASES, oracle_from_golden

pytestmark = pytest.mark.gpu

from test_gpu_parity import RANGE_CASES  # noqa: E402


def pair_engine(g, precision):
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision=precision)
    eng.set_template("pair")
    return eng


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", PAINN_CASES)
def test_pair_drift_vs_reference_golden(name, precision):
    g = load_golden(name)
    eng = pair_engine(g, precision)
    if name == "ambient_sparse":                            # still a symmetric graph: it has a pair layout
        assert eng.template_for(int(g["B"])) == "pair"
    if eng.template_for(int(g["B"])) != "pair":
        assert int(g["F"]) > 128                             # the only fixtures without one: F = 256
        pytest.skip("no pair-major layout at this width")
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        assert np.isfinite(got).all()
        assert rel_l2(got, g[f"drift_{i}"]) < DRIFT_TOL, (name, i)
        np.testing.assert_array_equal(eng.drift(g["x"], float(t), g["cond"]), got)            # deterministic


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", ["ambient_small", "ambient_sparse", "latent_multi", "ambient_full"])
def test_pair_stage_taps_vs_reference_intermediates(name, precision):
    """s, v, e after every message / update block through the pair-major row map (both directions of every pair)."""
    g = load_golden(name)
    eng, orc = pair_engine(g, precision), oracle_from_golden(g)
    B, A, F, L = int(g["B"]), int(g["A"]), int(g["F"]), int(g["L"])
    assert eng.template_for(B) == "pair"
    t = float(g["ts"][1])
    stages = [s for l in range(L) for s in ((1 + 2 * l, f"msg{l}"), (2 + 2 * l, f"upd{l}"))]
    try:
        for stage, tag in stages:
            eng.debug_tap(stage)
            eng.drift(g["x"], t, g["cond"])
            _, taps = orc.drift(g["x"], t, g["cond"], tap_stage=stage)
            s = eng.debug_read("s", B)
            v = eng.debug_read("v", B).transpose(0, 1, 3, 2)
            assert rel_l2(s, taps["s"]) < DRIFT_TOL, (tag, "s")
            assert rel_l2(v, taps["v"]) < DRIFT_TOL, (tag, "v")
            if f"im::s_{tag}" in g:
                assert rel_l2(s.reshape(B * A, F), g[f"im::s_{tag}"]) < DRIFT_TOL, (tag, "s golden")
                assert rel_l2(v.reshape(B * A, F, 3), g[f"im::v_{tag}"]) < DRIFT_TOL, (tag, "v golden")
            if tag.startswith("msg") and int(tag[3:]) < L - 1:
                e = eng.debug_read("e", B)
                assert rel_l2(e, taps["e"]) < DRIFT_TOL, (tag, "e")
                if f"im::e_{tag}" in g:
                    assert rel_l2(e.reshape(-1, F), g[f"im::e_{tag}"]) < DRIFT_TOL, (tag, "e golden")
    finally:
        eng.debug_tap(-1)
    assert rel_l2(eng.drift(g["x"], t, g["cond"]), g["drift_1"]) < DRIFT_TOL


@pytest.mark.parametrize("name", ["ambient_small", "ambient_a9", "ambient_full", "latent_multi"])
@pytest.mark.parametrize("scheme", ["euler", "heun"])
def test_pair_rollout_vs_reference_trajectory(name, scheme):
    g = load_golden(name)
    eng = pair_engine(g, "f16x2")
    path, nfe = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme=scheme, save_every=1)
    ref = g[f"traj_{scheme}"]
    assert nfe == (len(g["traj_grid"]) - 1) * (2 if scheme == "heun" else 1)
    assert rel_l2(path - path[0], ref - ref[0]) < 2e-5
    last, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme=scheme, save_every=0)
    np.testing.assert_array_equal(last[0], path[-1])


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", RANGE_CASES)
def test_pair_magnitude_edge_cases_vs_reference(name, precision):
    g = load_golden(name)
    eng = pair_engine(g, precision)
    orc = oracle_from_golden(g)
    for i, t in enumerate(g["ts"]):
        ref = g[f"drift_{i}"]
        floor = rel_l2(ref, orc.drift(g["x"], float(t), g["cond"], precision=64))       # the reference's own distance to exact arithmetic
        got = eng.drift(g["x"], float(t), g["cond"])
        assert np.isfinite(got).all()
        assert rel_l2(got, ref) < max(DRIFT_TOL, 3.0 * floor), (name, i, floor)


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("F,L,A,B,variant", [(64, 2, 18, 37, 0), (128, 2, 18, 131, 0), (32, 5, 3, 200, 2), (128, 1, 2, 5, 0), (128, 3, 25, 9, 1),
                                             (32, 2, -22, 9, 0)])
def test_pair_ragged_batches_and_templates_agree(F, L, A, B, variant, precision):
    """Batches that do not fill the last group of G molecules, tiny and sparse graphs; pair-major and directed rows evaluate the same
    sums in different orders: agreement to round-off with each other, to the drift bar with the fp64 oracle.  Stale workspace
    contents (first-touch accumulators, parked geometry of another batch) must not leak: a used engine equals a fresh one bit for bit."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    if A < 0:
        A = -A
        src, dst, et = syn.sparse_template(A, seed=1)
    else:
        src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
    x = syn.molecule_coords(B, A, seed=B)
    cond = [syn.ambient_cond(B, A), syn.latent_cond(B, A, 500.0), None][variant]
    kw = dict(temp_length=100.0 if variant == 0 else 75.0)
    mk = lambda: ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, precision=precision, **kw)
    eng = mk()
    orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, **kw)
    ref = orc.drift(x, 0.4, cond, precision=64)
    eng.set_template("pair")
    assert eng.template_for(B) == "pair"
    got = eng.drift(x, 0.4, cond)
    assert rel_l2(got, ref) < DRIFT_TOL
    eng.set_template("throughput")
    directed = eng.drift(x, 0.4, cond)
    assert rel_l2(got, directed) < 3e-6
    # the divergence entry points walk directed rows whatever is pinned
    eng.set_template("pair")
    b2, div = eng.drift_div(x[:3], 0.4, None if cond is None else cond[:3])
    _, rdiv = orc.drift_div(x[:3], 0.4, None if cond is None else cond[:3], precision=64)
    assert (np.abs(div - rdiv) < 2e-5 * (np.abs(rdiv) + 1.0)).all()
    # a used engine (other geometry, larger batch, directed run in between) against a fresh one
    x2 = (x + 0.2 * np.random.default_rng(1).standard_normal(x.shape)).astype(np.float32)
    fresh = mk()
    fresh.set_template("pair")
    np.testing.assert_array_equal(eng.drift(x2, 0.7, cond), fresh.drift(x2, 0.7, cond))
    # a slice evaluated alone (whole groups): same rows, same order -> bit for bit
    Gm = 8
    if B > 2 * Gm:
        part = eng.drift(x[:Gm], 0.4, None if cond is None else cond[:Gm])
        np.testing.assert_array_equal(part, got[:Gm])


def test_pair_race_screen_full_occupancy(monkeypatch):
    """The headline shape at full occupancy (8-wave workgroups, one per CU, replaced mid-launch): repeated launches agree bit for bit,
    the two matrix paths and the directed layout agree per molecule; six molecules against the fp64 oracle."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 16384
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    outs = {}
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        assert eng.template_for(B) == "pair"              # the automatic choice at this batch
        runs = [eng.drift(x, 0.5, cond) for _ in range(3)]
        np.testing.assert_array_equal(runs[1], runs[0])
        np.testing.assert_array_equal(runs[2], runs[0])
        outs[prec] = runs[0].reshape(B, -1)
        if prec == "f16x2":
            eng.set_template("throughput")
            outs["directed"] = eng.drift(x, 0.5, cond).reshape(B, -1)
        eng.close()
    ref = outs["f32"]
    scale = np.linalg.norm(ref, axis=1)
    for k in ("f16x2", "directed"):
        per_mol = np.linalg.norm(outs[k] - ref, axis=1) / scale
        assert per_mol.max() < 3e-5, f"{k}: {(per_mol >= 3e-5).sum()} molecules disagree, worst {per_mol.max():.2e}"
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    idx = np.r_[0:3, B - 3:B]
    assert rel_l2(outs["f16x2"][idx].reshape(-1, A, 3), orc.drift(x[idx], 0.5, cond[idx], precision=64)) < DRIFT_TOL


@pytest.mark.parametrize("layout", ["pair", "throughput"])
@pytest.mark.parametrize("poison", [float("nan"), 1e30])
def test_first_touch_accumulators_ignore_stale_contents(monkeypatch, layout, poison):
    """First-touch accumulators (csrc/painn_edge_kernel.hpp: acc_out; nothing zeroes dsacc / dvacc / cacc between layers or calls): with the
    accumulators POISONED before the evaluation (NaN, 1e30), at full occupancy and 18 atoms (every atom is touched from several row blocks,
    by different lanes of its wave), both layouts must return exactly what the zeroing path returns (TI_ZERO_ACC=1: memsets, the update
    kernel clears what it consumed, adds only).  The hardware property this rests on: no-return atomics of ONE wave to ONE address reach
    L2 in program order (exchange of block b before the adds of block b + 1), DESIGN.md 3.1."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 16384
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)

    def run(zeroing):
        if zeroing:
            monkeypatch.setenv("TI_ZERO_ACC", "1")
        else:
            monkeypatch.delenv("TI_ZERO_ACC", raising=False)
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16x2")
        monkeypatch.delenv("TI_ZERO_ACC", raising=False)
        eng.set_template(layout)
        outs = []
        for _ in range(2):                                   # the second call meets the first one's leftovers as well
            if not zeroing:
                eng.debug_poison(B, poison)
            outs.append(eng.drift(x, 0.5, cond))
        eng.close()
        return outs

    ref = run(True)
    got = run(False)
    assert np.isfinite(ref[0]).all()
    for g in got:
        np.testing.assert_array_equal(g, ref[0])
    np.testing.assert_array_equal(ref[1], ref[0])
