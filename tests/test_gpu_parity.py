"""GPU parity: the HIP path (through the C ABI) against the reference golden vectors and the CPU oracle.

Tolerances (stated once):
  DRIFT_TOL = 1e-5 rel-L2  -- the north-star bar for the fp32 drift vs the reference CPU path (BASELINE.json).
  Observed values are ~1-3e-6, the fp32 round-off floor of the reference itself (tests/test_oracle_golden.py).
Everything here needs a real MI355X: `pytest -m gpu`.
"""
import numpy as np
import pytest

from conftest import golden_weights, load_golden, pkg, rel_l2
from oracle import oracle

pytestmark = pytest.mark.gpu

DRIFT_TOL = 1e-5
PAINN_CASES = ["ambient_small", "ambient_sparse", "ambient_a9", "ambient_a25", "ambient_full", "ambient_ctor", "ambient_b1",
               "latent_multi", "latent_single", "latent_full", "latent_ctor"]


def engine_from_golden(g, device=0):
    ti = pkg()
    return ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                 g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                 device=device)


def oracle_from_golden(g):
    return oracle.PainnOracle(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                              g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"])


def test_mfma_lane_map_selftest():
    """ti_selftest: the MFMA lane maps, and the 8-instruction operand split (half-register writes) bit for bit against the plain
    arithmetic on 16.8 M values at full occupancy."""
    pkg().engine.selftest(0)


@pytest.mark.parametrize("precision", ["f32", "f16x2", "f16"])
def test_parked_geometry_is_per_evaluation(precision):
    """Layer 0 parks edge_dir and the distance encoding for the later layers of the SAME evaluation (painn_edge_kernel.hpp).  An
    engine that has evaluated one geometry, a larger batch, and a debug-tap run that stopped half way must give, bit for bit, what
    a fresh engine gives on the next geometry."""
    ti = pkg()
    g = load_golden("ambient_small")
    mk = lambda: ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                       g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                       precision=precision)
    rng = np.random.default_rng(5)
    x1 = g["x"]
    x2 = (x1 + 0.3 * rng.standard_normal(x1.shape)).astype(np.float32)
    t = float(g["ts"][1])
    fresh = mk().drift(x2, t, g["cond"])
    eng = mk()
    eng.drift(x1, t, g["cond"])
    reps = 5                                                   # a larger batch re-allocates the workspace (and the parked buffers)
    eng.drift(np.tile(x1, (reps, 1, 1)), t, np.tile(g["cond"], (reps,) + (1,) * (g["cond"].ndim - 1)))
    if precision != "f16":                                     # the storage mode has no taps
        eng.debug_tap(3)
        eng.drift(x1, t, g["cond"])
        eng.debug_tap(-1)
    again = eng.drift(x2, t, g["cond"])
    assert np.array_equal(again, fresh)
    assert not np.array_equal(again, eng.drift(x1, t, g["cond"]))


@pytest.mark.parametrize("name", PAINN_CASES)
def test_painn_drift_vs_reference_golden(name):
    g = load_golden(name)
    eng = engine_from_golden(g)
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        err = rel_l2(got, g[f"drift_{i}"])
        assert np.isfinite(got).all()
        assert err < DRIFT_TOL, (name, i, err)


@pytest.mark.parametrize("name", ["ambient_small", "ambient_sparse", "latent_multi"])
def test_painn_stage_taps_vs_oracle_and_golden(name):
    """s, v, e after the embed stage and after every message / update block."""
    g = load_golden(name)
    eng, orc = engine_from_golden(g), oracle_from_golden(g)
    B, A, F, L = int(g["B"]), int(g["A"]), int(g["F"]), int(g["L"])
    t = float(g["ts"][1])
    stages = [(0, "embed")] + [s for l in range(L) for s in ((1 + 2 * l, f"msg{l}"), (2 + 2 * l, f"upd{l}"))]
    try:
        for stage, tag in stages:
            eng.debug_tap(stage)
            eng.drift(g["x"], t, g["cond"])
            _, taps = orc.drift(g["x"], t, g["cond"], tap_stage=stage)
            s = eng.debug_read("s", B)
            assert rel_l2(s, taps["s"]) < DRIFT_TOL, (tag, "s")
            if tag == "embed":
                assert rel_l2(s.reshape(B * A, F), g["im::s_embed"]) < DRIFT_TOL
                continue
            v = eng.debug_read("v", B).transpose(0, 1, 3, 2)          # [B,A,3,F] -> reference [B,A,F,3]
            assert rel_l2(v, taps["v"]) < DRIFT_TOL, (tag, "v")
            assert rel_l2(v.reshape(B * A, F, 3), g[f"im::v_{tag}"]) < DRIFT_TOL, (tag, "v golden")
            assert rel_l2(s.reshape(B * A, F), g[f"im::s_{tag}"]) < DRIFT_TOL, (tag, "s golden")
            if tag.startswith("msg") and int(tag[3:]) < L - 1:       # the last message block's edge update is dead code
                e = eng.debug_read("e", B)
                assert rel_l2(e, taps["e"]) < DRIFT_TOL, (tag, "e")
                assert rel_l2(e.reshape(-1, F), g[f"im::e_{tag}"]) < DRIFT_TOL, (tag, "e golden")
    finally:
        eng.debug_tap(-1)
    # taps off again: a normal evaluation still matches
    assert rel_l2(eng.drift(g["x"], t, g["cond"]), g["drift_1"]) < DRIFT_TOL


@pytest.mark.parametrize("name", ["ambient_small", "ambient_a9", "ambient_full", "latent_multi"])
@pytest.mark.parametrize("scheme", ["euler", "heun"])
def test_painn_rollout_vs_reference_trajectory(name, scheme):
    g = load_golden(name)
    eng = engine_from_golden(g)
    path, nfe = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme=scheme, save_every=1)
    ref = g[f"traj_{scheme}"]
    assert path.shape == ref.shape
    assert nfe == (len(g["traj_grid"]) - 1) * (2 if scheme == "heun" else 1)
    np.testing.assert_array_equal(path[0], g["x"])
    assert rel_l2(path - path[0], ref - ref[0]) < 2e-5        # accumulated over the steps; per-step drift bar is 1e-5
    last, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme=scheme, save_every=0)
    np.testing.assert_array_equal(last[0], path[-1])           # deterministic: bit-identical re-run
    sub, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme=scheme, save_every=4)
    n = len(g["traj_grid"])
    rows = list(range(0, n, 4)) + ([n - 1] if (n - 1) % 4 else [])
    np.testing.assert_array_equal(sub, path[rows])


def test_em_eps0_is_euler_bit_for_bit_and_noise_matches_oracle():
    g = load_golden("ambient_small")
    eng, orc = engine_from_golden(g), oracle_from_golden(g)
    e, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="euler")
    em0, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="em", eps=0.0, seed=11)
    np.testing.assert_array_equal(e, em0)
    for com in (False, True):
        em, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="em", eps=0.05, seed=11, traj_offset=5, com_free_noise=com)
        ref, _ = orc.rollout(g["x"], g["cond"], g["traj_grid"], scheme="em", eps=0.05, seed=11, traj_offset=5, com_free_noise=int(com))
        assert rel_l2(em - em[0], ref - ref[0]) < 1e-4           # same Philox stream; libm differences in Box-Muller
        if com:     # one step: x_em - x_euler = sigma * (xi - COM(xi)) has no centre-of-mass component
            np.testing.assert_allclose((em[1] - e[1]).mean(axis=1), 0.0, atol=1e-6)
    # RNG is keyed by the global trajectory id: a shard starting at trajectory 1 reproduces rows 1.. of the full batch
    full, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="em", eps=0.05, seed=3)
    part, _ = eng.rollout(g["x"][1:], g["cond"][1:], g["traj_grid"], scheme="em", eps=0.05, seed=3, traj_offset=1)
    np.testing.assert_array_equal(full[:, 1:], part)


@pytest.mark.parametrize("precision", ["f32", "f16x2", "f16"])
@pytest.mark.parametrize("F,L,A,B,variant", [(64, 2, 18, 37, 0), (128, 2, 18, 130, 0), (256, 2, 25, 9, 1), (32, 5, 3, 200, 2), (128, 1, 2, 5, 0)])
def test_painn_ragged_batches_vs_oracle(F, L, A, B, variant, precision):
    """Batch sizes that are not multiples of the molecule-group size or of the 16-row tiles; every feature width (its own chunk size and
    superchunk depth per precision) on every matrix path.  The fp16 storage mode is held to its own bar."""
    DRIFT_TOL = 1e-2 if precision == "f16" else globals()["DRIFT_TOL"]
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
    x = syn.molecule_coords(B, A, seed=B)
    cond = [syn.ambient_cond(B, A), syn.latent_cond(B, A, 500.0), None][variant]
    kw = dict(temp_length=100.0 if variant == 0 else 75.0)
    eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, precision=precision, **kw)
    orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, **kw)
    got = eng.drift(x, 0.37, cond)
    assert rel_l2(got, orc.drift(x, 0.37, cond)) < DRIFT_TOL
    # the same engine re-used with a smaller and a larger batch (workspace regrowth)
    for b2 in (1, B + 77):
        x2 = syn.molecule_coords(b2, A, seed=b2)
        c2 = [syn.ambient_cond(b2, A), syn.latent_cond(b2, A, 500.0), None][variant]
        assert rel_l2(eng.drift(x2, 0.9, c2), orc.drift(x2, 0.9, c2)) < DRIFT_TOL


def test_painn_large_batch_properties():
    """BASELINE-size features (F=128, L=5, A=18) on a batch the oracle cannot finish quickly: size-independent
    properties of the architecture -- rotation equivariance, translation invariance, independence of molecules, determinism."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 4096
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    b = eng.drift(x, 0.5, cond)
    assert np.isfinite(b).all()
    np.testing.assert_array_equal(b, eng.drift(x, 0.5, cond))                       # deterministic
    q, _ = np.linalg.qr(np.random.RandomState(1).standard_normal((3, 3)))
    q = (q * np.sign(np.linalg.det(q))).astype(np.float32)                          # proper rotation (the cross-product channel is chiral)
    assert rel_l2(eng.drift(x @ q.T, 0.5, cond), b @ q.T) < DRIFT_TOL
    assert rel_l2(eng.drift(x + np.float32([0.3, -0.2, 0.1]), 0.5, cond), b) < DRIFT_TOL
    perm = np.random.RandomState(2).permutation(B)
    # molecules are independent; the position inside a molecule group only changes the order of a few partial sums
    assert rel_l2(eng.drift(x[perm], 0.5, cond[perm]), b[perm]) < 5e-6
    # a slice of the batch evaluated on its own (group/tile padding does not leak): bit for bit with the same edge template;
    # a small batch alone switches to the latency template (other row blocking, same sums in another order)
    assert rel_l2(eng.drift(x[:131], 0.5, cond[:131]), b[:131]) < 2e-6
    # (131 molecules = 65.5 directed groups of two; the pair-major layout walks groups of four: 132)
    for layout, n in (("throughput", 131), ("pair", 132)):
        eng.set_template(layout)
        np.testing.assert_array_equal(eng.drift(x[:n], 0.5, cond[:n]), eng.drift(x, 0.5, cond)[:n])
    eng.set_template("auto")
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    idx = np.r_[0:3, B - 3:B]
    assert rel_l2(b[idx], orc.drift(x[idx], 0.5, cond[idx])) < DRIFT_TOL


@pytest.mark.parametrize("template,B", [("throughput", 16384), ("latency", 6000)])
def test_painn_race_screen_full_occupancy(template, B, monkeypatch):
    """Every CU holds two workgroups and workgroups are replaced mid-launch (1366 workgroups, 512 resident): the weight
    stream's LDS-DMA / barrier protocol is exercised under memory load, where an early fragment read shows up as a few
    wrong molecules per launch (seen during development at ~1 in 4096; invisible at the small parity sizes).  The f32 and
    split-fp16 builds are independent instruction streams, so per-molecule agreement of repeated launches of both is the
    screen; the absolute check against the oracle is the small-batch parity tests' job."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    monkeypatch.setenv("TI_TEMPLATE", template)          # both edge-row layouts (ti_internal.hpp) under full occupancy
    F, L, A = 128, 5, 18
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=0), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    outs = []
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        outs += [eng.drift(x, 0.5, cond).reshape(B, -1) for _ in range(3)]
        eng.close()
    ref = outs[0]
    scale = np.linalg.norm(ref, axis=1)
    for o in outs[1:]:
        per_mol = np.linalg.norm(o - ref, axis=1) / scale
        assert per_mol.max() < 3e-5, f"{(per_mol >= 3e-5).sum()} molecules disagree, worst {per_mol.max():.2e}"


# ----------------------------------------------------------------------------------------------------------- adw
def adw_engine(g):
    ti = pkg()
    H, nl = int(g["hidden"]), int(g["num_layers"])
    spec = ti.weights.adw_param_spec(H, nl)
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")} or ti.synthetic.adw_state_dict(H, nl, int(g["seed"]))
    flat = ti.weights.flatten_state_dict(sd, spec, dtype=np.float64)
    return ti.engine.AdwEngine(H, nl, flat), oracle.AdwOracle(H, nl, flat)


@pytest.mark.parametrize("name", ["adw_h256", "adw_ctor_h64"])
def test_adw_drift_and_rollout_vs_reference(name):
    g = load_golden(name)
    eng, orc = adw_engine(g)
    for tag in ("", "_var"):
        b0, b1 = g["beta0" + tag].astype(np.float32), g["beta1" + tag].astype(np.float32)
        for i, t in enumerate(g["ts"]):
            got = eng.drift(g["x"], float(t), b0, b1)
            assert rel_l2(got, g[f"drift{tag}_{i}"]) < DRIFT_TOL, (tag, i)          # fp32 device vs fp64 reference
    b0, b1 = g["beta0"].astype(np.float32), g["beta1"].astype(np.float32)
    for scheme in ("euler", "heun"):
        path, nfe = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme=scheme)
        ref = g[f"traj_{scheme}"]
        assert path.shape == ref.shape
        assert rel_l2(path - path[0], ref - ref[0]) < 2e-5
    e, _ = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme="euler")
    em0, _ = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme="em", eps=0.0, seed=5)
    np.testing.assert_array_equal(e, em0)
    em, _ = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme="em", eps=0.1, seed=5)
    ref, _ = orc.rollout(g["x"].astype(np.float64), b0, b1, g["traj_grid"], scheme="em", eps=0.1, seed=5)
    assert rel_l2(em, ref) < 1e-4


def test_adw_large_batch_vs_oracle():
    ti = pkg()
    flat = ti.weights.flatten_state_dict(ti.synthetic.adw_state_dict(256, 5, 0), ti.weights.adw_param_spec(256, 5), dtype=np.float64)
    eng, orc = ti.engine.AdwEngine(256, 5, flat), oracle.AdwOracle(256, 5, flat)
    B = 10_007
    x = ti.synthetic.adw_x0(B, 3)
    b0, b1 = np.full(B, 1.0, np.float32), np.full(B, 1.25, np.float32)
    assert rel_l2(eng.drift(x, 0.4, b0, b1), orc.drift(x.astype(np.float64), 0.4, b0, b1)) < DRIFT_TOL


# ------------------------------------------------------------------------------------------------- device buffers
def test_device_resident_buffers_match_host_path():
    torch = pytest.importorskip("torch")
    g = load_golden("ambient_a9")
    eng = engine_from_golden(g)
    host, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="heun")
    xd, cd = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["cond"]).cuda()
    dev, _ = eng.rollout(xd, cd, g["traj_grid"], scheme="heun")
    assert dev.is_cuda
    np.testing.assert_array_equal(dev.cpu().numpy(), host)
    np.testing.assert_array_equal(eng.drift(xd, 0.25, cd).cpu().numpy(), eng.drift(g["x"], 0.25, g["cond"]))


# ------------------------------------------------------------------------------------------------- split-fp16 mode
@pytest.mark.parametrize("name", PAINN_CASES)
def test_split_fp16_precision_mode_meets_the_same_bar(name):
    """precision='f16x2': the message MLPs' products on the fp16 matrix rate, operands split hi + 2^-11 lo.  Same bar
    as the f32-MFMA path against the reference goldens; the measured error is reported for DESIGN.md."""
    g = load_golden(name)
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision="f16x2")
    ref32 = engine_from_golden(g)
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        err, err32 = rel_l2(got, g[f"drift_{i}"]), rel_l2(ref32.drift(g["x"], float(t), g["cond"]), g[f"drift_{i}"])
        print(f"\n[split-fp16] {name} t={float(t):.2f}: rel-L2 vs reference {err:.2e} (f32-MFMA path {err32:.2e})")
        assert np.isfinite(got).all() and err < DRIFT_TOL, (name, i, err)


# ------------------------------------------------------------------------------------- adw divergence / dlogp, split mode
@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", ["adw_h256", "adw_ctor_h64"])
def test_adw_divergence_and_dlogp(name, precision):
    """Exact divergence by forward-mode differentiation of `net` against the reference's autograd value
    (ODEWrapper.compute_divergence, golden key negdiv* = -div * 1e-2 at t = 0.3), and the integrated dlogp state of
    StandardIntegrator(return_dlogp=True) against the oracle's fp64 rollout."""
    ti = pkg()
    g = load_golden(name)
    H, nl = int(g["hidden"]), int(g["num_layers"])
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")} or ti.synthetic.adw_state_dict(H, nl, int(g["seed"]))
    flat = ti.weights.flatten_state_dict(sd, ti.weights.adw_param_spec(H, nl), dtype=np.float64)
    eng, orc = ti.engine.AdwEngine(H, nl, flat, precision=precision), oracle.AdwOracle(H, nl, flat)
    for tag in ("", "_var"):
        b0, b1 = g["beta0" + tag].astype(np.float32), g["beta1" + tag].astype(np.float32)
        b, div = eng.drift(g["x"], 0.3, b0, b1, return_div=True)
        assert rel_l2(-div * 1e-2, g[f"negdiv{tag}_1"].ravel()) < DRIFT_TOL, (tag, precision)
        for i, t in enumerate(g["ts"]):
            assert rel_l2(eng.drift(g["x"], float(t), b0, b1), g[f"drift{tag}_{i}"]) < DRIFT_TOL, (tag, i, precision)
    b0, b1 = g["beta0"].astype(np.float32), g["beta1"].astype(np.float32)
    for scheme in ("euler", "heun"):
        x, dl, nfe = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme=scheme, return_dlogp=True)
        xr, dlr, _ = orc.rollout(g["x"].astype(np.float64), b0, b1, g["traj_grid"], scheme=scheme, return_dlogp=True)
        assert x.shape == dl.shape == xr.shape and np.all(dl[0] == 0)
        assert rel_l2(x, xr) < DRIFT_TOL and rel_l2(dl, dlr) < 2e-5, (scheme, precision)
        x2, _ = eng.rollout(g["x"], b0, b1, g["traj_grid"], scheme=scheme)
        np.testing.assert_array_equal(x, x2)                         # the dlogp state does not perturb the trajectory


def test_adw_standard_integrator_with_dlogp_like_the_shipped_config():
    """adw/config/settings.json ships return_dlogp = 1: rollout returns (x [n_step,B,1], dlogp*1e2 [n_step,B,1])."""
    torch = pytest.importorskip("torch")
    ti = pkg()
    g = load_golden("adw_ctor_h64")
    net = ti.thermo.adw.FCNetMultiBeta(1, 1, int(g["hidden"]), int(g["num_layers"]))
    net.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd::")})
    x0s = torch.from_numpy(g["x"])[:, None]
    beta0s = torch.from_numpy(g["beta0"])[:, None]
    beta1s = torch.ones_like(beta0s) * 1.25
    n_step = len(g["traj_grid"])
    integ = ti.thermo.adw.StandardIntegrator(b=net, method="heun", rtol=1e-4, atol=1e-4, n_step=n_step, return_dlogp=1)
    sample, dlogp = integ.rollout(x0s, beta0s=beta0s, beta1s=beta1s)
    assert tuple(sample.shape) == tuple(dlogp.shape) == (n_step, len(g["x"]), 1)
    assert rel_l2(sample.numpy()[:, :, 0], g["traj_heun"]) < 1e-5
    b, negdiv = ti.thermo.adw.ODEWrapper(net, return_dlogp=True)(0.3, (x0s, None), None, beta0s, beta1s)
    assert rel_l2(negdiv.numpy().ravel(), g["negdiv_1"].ravel()) < 1e-5


@pytest.mark.parametrize("F,L,A,B,variant,precision", [(128, 2, 18, 50, 0, "f32"), (128, 2, 18, 50, 0, "f16x2"), (64, 2, 25, 7, 1, "f32"),
                                                       (32, 3, 9, 33, 2, "f32"), (32, 2, 4, 5, 0, "f32"), (32, 2, -22, 9, 0, "f32")])
def test_edge_templates_agree(F, L, A, B, variant, precision, monkeypatch):
    """The two edge templates (ti_internal.hpp: throughput = G molecules per wave, latency = one molecule cut into parts of
    destination atoms, one wave each) evaluate the same sums in different blockings: drift, per-stage state and divergence agree
    with each other to round-off and with the oracle to the drift bar.  A = 4 has too few rows for a second template."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    if A < 0:                                           # a sparse graph: uneven in-degrees (1 .. ~10), parts of unequal length
        A = -A
        src, dst, et = syn.sparse_template(A, seed=1)
    else:
        src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(variant, F, L, 25, seed=F + A), W.painn_param_spec(variant, F, L, 25))
    x = syn.molecule_coords(B, A, seed=B)
    cond = [syn.ambient_cond(B, A), syn.latent_cond(B, A, 500.0), None][variant]
    kw = dict(temp_length=100.0 if variant == 0 else 75.0)
    eng = ti.engine.PainnEngine(variant, F, L, A, src, dst, et, np.arange(A), flat, precision=precision, **kw)
    orc = oracle.PainnOracle(variant, F, L, A, src, dst, et, np.arange(A), flat, **kw)
    ref, rdiv = orc.drift_div(x, 0.4, cond, precision=64)
    res = {}
    for mode in ("throughput", "latency"):
        monkeypatch.setenv("TI_TEMPLATE", mode)
        b, div = eng.drift_div(x, 0.4, cond)
        assert rel_l2(b, ref) < DRIFT_TOL, mode
        assert (np.abs(div - rdiv) < 2e-5 * (np.abs(rdiv) + 1.0)).all(), mode
        np.testing.assert_array_equal(eng.drift(x, 0.4, cond), b)
        eng.debug_tap(1)                                     # after the first message block: the e rows go through the row map
        try:
            eng.drift(x, 0.4, cond)
            _, taps = orc.drift(x, 0.4, cond, tap_stage=1)
            if L > 1:
                assert rel_l2(eng.debug_read("e", B), taps["e"]) < DRIFT_TOL, mode
            assert rel_l2(eng.debug_read("s", B), taps["s"]) < DRIFT_TOL, mode
        finally:
            eng.debug_tap(-1)
        path, dl, _ = eng.rollout_dlogp(x, cond, np.linspace(0, 1, 3).astype(np.float32), scheme="heun", save_every=0)
        res[mode] = (b, div, path, dl)
    assert rel_l2(res["latency"][0], res["throughput"][0]) < 2e-6
    assert rel_l2(res["latency"][2], res["throughput"][2]) < 2e-6
    assert np.abs(res["latency"][3] - res["throughput"][3]).max() < 2e-5 * (np.abs(res["throughput"][3]).max() + 1.0)


@pytest.mark.parametrize("F", [32, 64])
def test_wide_workgroups_at_small_feature_widths(F, monkeypatch):
    """Launches with >= 2048 molecule groups use the 8-wave edge kernels (painn_kernels.hip: launch_edge); the headline shape
    covers F = 128 (race screen above), this covers the F = 32 / 64 builds, whose weight superchunks stay two chunks deep."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    monkeypatch.setenv("TI_TEMPLATE", "throughput")
    L, A, B = 2, 6, 8 * 2048 + 5                       # E = 30 -> G = 8 molecules per group (240 rows, no padding): 2049 groups
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=F), W.painn_param_spec(0, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=1), syn.ambient_cond(B, A)
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    idx = np.r_[0:40, B - 40:B]
    ref = orc.drift(x[idx], 0.3, cond[idx])
    outs = {}
    for prec in ("f32", "f16x2"):
        eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=prec)
        b = eng.drift(x, 0.3, cond)
        assert rel_l2(b[idx], ref) < DRIFT_TOL, prec
        np.testing.assert_array_equal(eng.drift(x, 0.3, cond), b)
        # the same molecules through the narrow build (a batch below the threshold): same sums, same order -> bit for bit
        np.testing.assert_array_equal(eng.drift(x[:800], 0.3, cond[:800]), b[:800])
        outs[prec] = b.reshape(B, -1)
        eng.close()
    per_mol = np.linalg.norm(outs["f16x2"] - outs["f32"], axis=1) / np.linalg.norm(outs["f32"], axis=1)
    assert per_mol.max() < 3e-5


RANGE_CASES = ["range_big", "range_big_f128", "range_tiny", "range_tiny_f128", "range_close", "range_latent_big"]


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", RANGE_CASES)
def test_magnitude_edge_cases_vs_reference(name, precision):
    """Reference fixtures whose un-normalised streams (s, e, v, |Vv|: cpainn.py:306-308, 363-374) reach 1e3 .. 1e6 or only
    1e-7 .. 1e-9, and an edge of length 1e-4.  Both matrix paths must meet the drift bar: the split-fp16 path scales every such
    operand row by a power of two before the hi/lo split (mfma_chain.hpp: Opnd::set_scaled), so neither the fp16 range
    (65504) nor its subnormals limit it.  The bar is relative to the reference's own fp32 round-off on these cases."""
    g = load_golden(name)
    ti = pkg()
    if name.startswith("range_big") or name.startswith("range_tiny"):
        im = {k[4:]: float(np.abs(v).max()) for k, v in g.items() if k.startswith("im::")}
        if name.startswith("range_big"):
            assert im["e_msg0"] > 5e3 and im["s_msg0"] > 3e3 and im["v_upd1"] > 1e5          # beyond fp16's 65504 in layer 2
        else:
            assert im["e_msg0"] < 1e-6 and im["s_msg0"] < 1e-6 and im["v_msg0"] < 1e-7       # fp16-subnormal territory
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision=precision)
    orc = oracle_from_golden(g)
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        assert np.isfinite(got).all(), (name, precision)
        err = rel_l2(got, g[f"drift_{i}"])
        exact = orc.drift(g["x"], float(t), g["cond"], precision=64)
        floor = rel_l2(g[f"drift_{i}"], exact)             # the reference's own distance to exact arithmetic
        assert err < max(DRIFT_TOL, 3 * floor), (name, precision, i, err, floor)


LNAFF_CASES = ["lnaff_1em5_f32", "lnaff_1em5_f128", "lnaff_1em3_f32", "lnaff_1em3_f128", "lnaff_1e3_f32", "lnaff_1e3_f128",
               "lnaff_harsh_f32", "lnaff_harsh_f128", "lnaff_zero_w_f32"]


def check_lnaff_magnitudes(name, g):
    """The fixture really drives the hidden activations (SiLU outputs inside the message / update MLPs) where it says."""
    h = g["im::hidden_absmax"]
    if "zero_w" in name:
        return                                              # (a weight-scale case, not an activation-magnitude one)
    if "1em5" in name or "harsh" in name:
        assert h.max() < 6.1e-5                             # every hidden row below fp16's smallest normal number
    elif "1em3" in name:
        assert 1e-4 < h.max() < 1e-2
    else:
        assert h.max() > 2e3


@pytest.mark.parametrize("template", ["throughput", "pair"])
@pytest.mark.parametrize("precision", ["f32", "f16x2"])
@pytest.mark.parametrize("name", LNAFF_CASES)
def test_layernorm_affine_magnitudes_vs_reference(name, precision, template):
    """Reference fixtures whose message / update LayerNorm affines (embedding.py:27-35) are rescaled by 1e-5, 1e-3 and 1e3: the
    hidden activations -- the operands the split-fp16 path feeds to the next matrix product -- sit in fp16's subnormal range, two
    binades above it, or at 4e3.  `harsh` also shrinks the next Linear's bias, so nothing swamps what the product loses.  Bar as
    for the other magnitude cases: max(1e-5, 3 x the reference's own distance to exact arithmetic), both matrix paths, directed
    and pair-major message kernels."""
    g = load_golden(name)
    check_lnaff_magnitudes(name, g)
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision=precision)
    eng.set_template(template)
    orc = oracle_from_golden(g)
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        assert np.isfinite(got).all(), (name, precision)
        err = rel_l2(got, g[f"drift_{i}"])
        floor = rel_l2(g[f"drift_{i}"], orc.drift(g["x"], float(t), g["cond"], precision=64))
        assert err < max(DRIFT_TOL, 3 * floor), (name, precision, template, i, err, floor)


def test_f16x2_refuses_weights_beyond_the_fp16_range():
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    flat = W.flatten_state_dict(syn.painn_state_dict(0, 32, 1, 25, 0), W.painn_param_spec(0, 32, 1, 25)).copy()
    flat[100] = 1e5
    with pytest.raises(ti._lib.TiError) as ei:
        ti.engine.PainnEngine(0, 32, 1, 4, *syn.fully_connected_template(4), np.arange(4), flat, precision="f16x2")
    assert "65504" in str(ei.value)
    ti.engine.PainnEngine(0, 32, 1, 4, *syn.fully_connected_template(4), np.arange(4), flat, precision="f32").close()


# ------------------------------------------------------------------------------------------------- fp16 storage mode
F16_TOL = 1e-2          # a separately labelled precision (BASELINE.json configs[4]): NOT the 1e-5 parity bar of the two fp32-grade paths


@pytest.mark.parametrize("name", PAINN_CASES)
def test_fp16_storage_mode_error_is_bounded_and_reported(name):
    """precision='f16': node / edge state in HBM as fp16, ONE fp16 product per k-step (fp32 accumulation, LayerNorm, sums).  The
    drift must stay within 1e-2 rel-L2 of the reference goldens; the measured error is printed for DESIGN.md."""
    g = load_golden(name)
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision="f16")
    for i, t in enumerate(g["ts"]):
        got = eng.drift(g["x"], float(t), g["cond"])
        err = rel_l2(got, g[f"drift_{i}"])
        print(f"\n[fp16 storage] {name} t={float(t):.2f}: rel-L2 vs reference {err:.2e}")
        assert np.isfinite(got).all() and err < F16_TOL, (name, i, err)
    again = eng.drift(g["x"], float(g["ts"][0]), g["cond"])
    np.testing.assert_array_equal(again, eng.drift(g["x"], float(g["ts"][0]), g["cond"]))      # still bit-reproducible


def test_fp16_storage_mode_rollout_and_both_templates():
    """An Euler rollout in the storage mode tracks the reference trajectory fixture to the same bar, the two edge-row layouts
    agree to the mode's own round-off, and a molecule's result does not depend on the rest of the batch (bit for bit, within a layout)."""
    g = load_golden("ambient_full")
    ti = pkg()
    mk = lambda: ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                       g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                       precision="f16")
    eng = mk()
    traj, _ = eng.rollout(g["x"], g["cond"], g["traj_grid"], scheme="euler")
    err = rel_l2(traj - traj[0], g["traj_euler"] - g["traj_euler"][0])
    print(f"\n[fp16 storage] ambient_full Euler displacement path: rel-L2 vs reference {err:.2e}")
    assert err < F16_TOL
    B = 37
    x = np.concatenate([g["x"]] * (B // g["x"].shape[0] + 1))[:B] * np.linspace(0.9, 1.1, B, dtype=np.float32)[:, None, None]
    cond = np.concatenate([g["cond"]] * (B // g["cond"].shape[0] + 1))[:B]
    outs = {}
    for which in ("throughput", "latency"):
        eng.set_template(which)
        outs[which] = eng.drift(x, 0.3, cond)
        np.testing.assert_array_equal(eng.drift(x[4:6], 0.3, cond[4:6]), outs[which][4:6])          # a whole group of the throughput layout
    assert rel_l2(outs["latency"], outs["throughput"]) < 2e-3


def test_fp16_storage_mode_refuses_what_it_does_not_cover():
    """Divergence / dlogp / tangents / stage taps are fp32-grade products only: the storage mode refuses them loudly."""
    g = load_golden("ambient_small")
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision="f16")
    with pytest.raises(ti._lib.TiError):
        eng.drift_div(g["x"], 0.25, g["cond"])
    with pytest.raises(ti._lib.TiError):
        eng.rollout_dlogp(g["x"], g["cond"], g["traj_grid"], scheme="euler")
    with pytest.raises(ti._lib.TiError):
        eng.jvp(g["x"], np.ones_like(g["x"]), 0.25, g["cond"])


def test_fp16_storage_mode_reports_state_overflow():
    """A state beyond the fp16 range (the range_big fixture drives v to 1e5 .. 1e6) cannot be stored in this mode: the rollout must
    come back with TI_E_NAN, not with numbers."""
    g = load_golden("range_big_f128")
    ti = pkg()
    eng = ti.engine.PainnEngine(int(g["variant"]), int(g["F"]), int(g["L"]), int(g["A"]), g["edge_src"], g["edge_dst"], g["edge_type"],
                                g["atom_ids"], golden_weights(g), temp_length=float(g["temp_length"]), temperatures=g["temperatures"],
                                precision="f16")
    with pytest.raises(ti._lib.TiError) as ei:
        eng.rollout(g["x"], g["cond"], ti.engine.time_grid(0.0, 1.0, 3), scheme="euler", save_every=0)
    assert ei.value.code == ti._lib.TI_E_NAN


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_atom_without_incoming_edges_keeps_the_zeroing_path(precision):
    """The accumulators are normally never zeroed: an atom's first row block REPLACES their contents (first-touch flag in the slot
    table).  That needs every atom to receive messages; a graph with an atom nobody sends to falls back to explicit zeroing.  Both
    must agree with the oracle, also on the second evaluation (stale accumulator contents) and in a multi-step rollout."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    A, F, L, B = 6, 64, 3, 70
    src, dst, et = syn.fully_connected_template(A)
    keep = dst != A - 1                                   # nobody sends to the last atom; it still sends to everyone
    src, dst, et = src[keep], dst[keep], et[keep]
    flat = W.flatten_state_dict(syn.painn_state_dict(0, F, L, 25, seed=3), W.painn_param_spec(0, F, L, 25))
    eng = ti.engine.PainnEngine(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision=precision)
    orc = oracle.PainnOracle(0, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    x, cond = syn.molecule_coords(B, A, seed=1), syn.ambient_cond(B, A)
    for t in (0.2, 0.7, 0.2):
        assert rel_l2(eng.drift(x, t, cond), orc.drift(x, t, cond)) < DRIFT_TOL
    grid = ti.engine.time_grid(0.0, 1.0, 4)
    got, _ = eng.rollout(x, cond, grid, scheme="euler", save_every=0)
    want, _ = orc.rollout(x, cond, grid, scheme="euler", save_every=0)
    assert rel_l2(got - x, want - x) < 2e-5


# ------------------------------------------------------------------------------------------- BASELINE.json configs at full batch
def _full_batch_checks(eng, orc, x, cond, *, scheme, eps, tol, template):
    """Property checks at a BASELINE.json batch size (the parity proper is the fixture tests' job): finite, bit-identical re-run, a
    132-molecule slice evaluated alone under the same pinned layout equals the slice of the big run bit for bit, six molecules
    against the CPU oracle."""
    ti = pkg()
    B = x.shape[0]
    eng.set_template(template)
    assert eng.template_for(B) == template
    grid = ti.engine.time_grid(0.0, 1.0, 1001)[:2]
    kw = dict(scheme=scheme, eps=eps, seed=3, save_every=0)
    a, nfe = eng.rollout(x, cond, grid, **kw)
    assert nfe == 1 and np.isfinite(a).all()
    b, _ = eng.rollout(x, cond, grid, **kw)
    np.testing.assert_array_equal(a, b)
    part, _ = eng.rollout(x[:132], None if cond is None else cond[:132], grid, **kw)
    np.testing.assert_array_equal(part[0], a[0, :132])
    idx = np.r_[0:3, B - 3:B]
    d = eng.drift(x[idx], 0.5, None if cond is None else cond[idx])
    assert rel_l2(d, orc.drift(x[idx], 0.5, None if cond is None else cond[idx], precision=64)) < tol


def test_config3_latent_sampler_full_batch():
    """BASELINE.json configs[2]: mdqm9 latent sampler, 65 536 molecules x 18 atoms, F = 128, L = 5, one ODE (Euler) step."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 65536
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.LATENT_MULTI, F, L, 25, 9), W.painn_param_spec(W.LATENT_MULTI, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=2, sigma=1.0), syn.latent_cond(B, A, 800.0)
    eng = ti.engine.PainnEngine(W.LATENT_MULTI, F, L, A, src, dst, et, np.arange(A), flat, temp_length=75.0, precision="f16x2")
    orc = oracle.PainnOracle(W.LATENT_MULTI, F, L, A, src, dst, et, np.arange(A), flat, temp_length=75.0)
    _full_batch_checks(eng, orc, x, cond, scheme="euler", eps=0.0, tol=DRIFT_TOL, template=eng.template_for(B))


def test_config4_ambient_sampler_full_batch():
    """BASELINE.json configs[3] (the bench shape): 65 536 ambient molecules, Euler-Maruyama step over the 6-rung ladder, the layout
    the library picks for this batch by itself."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 65536
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.AMBIENT, F, L, 25, 0), W.painn_param_spec(W.AMBIENT, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=0), syn.ambient_cond(B, A)
    eng = ti.engine.PainnEngine(W.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16x2")
    orc = oracle.PainnOracle(W.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    _full_batch_checks(eng, orc, x, cond, scheme="em", eps=0.01, tol=DRIFT_TOL, template=eng.template_for(B))


def test_config5_fp16_storage_full_share():
    """BASELINE.json configs[4]: one GPU's share (131 072 molecules) of the 1 048 576-molecule fp16-node-feature run, one EM step in
    the separately labelled storage mode (bar 1e-2, tests above)."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    F, L, A, B = 128, 5, 18, 131072
    src, dst, et = syn.fully_connected_template(A)
    flat = W.flatten_state_dict(syn.painn_state_dict(W.AMBIENT, F, L, 25, 0), W.painn_param_spec(W.AMBIENT, F, L, 25))
    x, cond = syn.molecule_coords(B, A, seed=4), syn.ambient_cond(B, A)
    eng = ti.engine.PainnEngine(W.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0, precision="f16")
    orc = oracle.PainnOracle(W.AMBIENT, F, L, A, src, dst, et, np.arange(A), flat, temp_length=100.0)
    _full_batch_checks(eng, orc, x, cond, scheme="em", eps=0.01, tol=1e-2, template="throughput")


def test_config2_adw_full_batch():
    """BASELINE.json configs[1]: 262 144 particles, Euler-Maruyama steps of the 1000-step grid (four of them), HIP MLP drift."""
    ti = pkg()
    syn, W = ti.synthetic, ti.weights
    B = 262144
    flat = W.flatten_state_dict(syn.adw_state_dict(256, 5, 0), W.adw_param_spec(256, 5), dtype=np.float64)
    eng, orc = ti.engine.AdwEngine(256, 5, flat, precision="f16x2"), oracle.AdwOracle(256, 5, flat)
    x = syn.adw_x0(B, 0)
    b0, b1 = np.full(B, 1.0, np.float32), np.full(B, 1.25, np.float32)
    grid = ti.engine.time_grid(0.0, 1.0, 1001)[:5]
    kw = dict(scheme="em", eps=0.05, seed=5, save_every=0)
    a, nfe = eng.rollout(x, b0, b1, grid, **kw)
    assert nfe == 4 and np.isfinite(a).all()
    b, _ = eng.rollout(x, b0, b1, grid, **kw)
    np.testing.assert_array_equal(a, b)
    part, _ = eng.rollout(x[:1000], b0[:1000], b1[:1000], grid, **kw)         # noise is keyed by the global particle index
    np.testing.assert_array_equal(part[0], a[0, :1000])
    idx = np.r_[0:512, B - 512:B]
    ref, _ = orc.rollout(x[idx].astype(np.float64), b0[idx], b1[idx], grid, scheme="euler")
    got, _ = eng.rollout(x[idx], b0[idx], b1[idx], grid, scheme="euler", save_every=0)
    assert rel_l2(got[0], ref[-1]) < 1e-5
