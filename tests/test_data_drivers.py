"""Graph templates, dataset layout and driver plumbing: CPU checks on synthetic files, and one
GPU end-to-end run of the drivers."""
import json
import os
import types

import numpy as np
import pytest

from conftest import load_golden, pkg, rel_l2

MOLBLOCK = """ethanol-like
  test

  4  3  0  0  0  0  0  0  0  0999 V2000
    0.0000    0.0000    0.0000 C   0  0
    1.5000    0.0000    0.0000 C   0  0
    2.1000    1.2000    0.0000 O   0  0
    2.9000    1.1000    0.5000 H   0  0
  1  2  1  0
  2  3  2  0
  3  4  4  0
M  END
"""


def test_edge_template_matches_reference_construction():
    ti = pkg()
    d, syn = ti.data, ti.synthetic
    x = syn.molecule_coords(1, 6, 3)[0]
    # fully connected radius graph + chain bonds == the synthetic template used everywhere else (sorted by (src, dst))
    chain = np.arange(5)
    bond_index = np.stack([np.concatenate([chain, chain + 1]), np.concatenate([chain + 1, chain])])
    src, dst, et = d.build_edge_template(x, 1000.0, bond_index, np.ones(10, np.int64))
    rs, rd, rt = syn.fully_connected_template(6)
    np.testing.assert_array_equal(src, rs); np.testing.assert_array_equal(dst, rd); np.testing.assert_array_equal(et, rt)
    # finite cutoff: symmetric, no self loops, bond type wins over radius type 0 (coalesce max), bonds outside the cutoff stay
    x = np.array([[0, 0, 0], [1, 0, 0], [2.5, 0, 0], [9, 0, 0]], np.float32)
    bi, bo = d.bonds_from_molblock(MOLBLOCK)
    np.testing.assert_array_equal(bi, [[0, 1, 2, 1, 2, 3], [1, 2, 3, 0, 1, 2]])
    np.testing.assert_array_equal(bo, [1, 2, 1, 1, 2, 1])          # aromatic (4) -> 1
    src, dst, et = d.build_edge_template(x, 1.6, bi, bo)
    got = {(int(a), int(b)): int(t) for a, b, t in zip(src, dst, et)}
    assert got == {(0, 1): 1, (1, 0): 1, (1, 2): 2, (2, 1): 2, (2, 3): 1, (3, 2): 1}
    assert list(zip(src, dst)) == sorted(zip(src, dst))
    s2, d2, t2 = d.coalesce_max([1, 0, 1, 1], [0, 1, 0, 2], [0, 3, 2, 1])
    assert list(zip(s2, d2, t2)) == [(0, 1, 3), (1, 0, 2), (1, 2, 1)]


def test_dataset_layout_and_batches(tmp_path):
    ti = pkg()
    d = ti.data
    A, n = 5, 12
    rs = np.random.RandomState(0)
    traj = rs.standard_normal((8, n, A, 3)) + 3.0                  # [temperature, frame, atom, xyz], not centred
    os.makedirs(tmp_path / "test")
    np.save(tmp_path / "test" / "00031.npy", traj)
    ds = d.MDQM9SamplerDataset("00031.npy", str(tmp_path), "test", T0=1000, T1=300, scale=True, cutoff=1000)
    assert len(ds) == n
    ref = traj[7] - traj[7].mean(axis=1, keepdims=True)
    np.testing.assert_allclose(ds.data, ref * d.SCALING_FACTOR_31, rtol=1e-6)
    batches = list(ds.batches(5, shuffle=True, seed=1))
    assert [b.x0.shape[0] for b in batches] == [25, 25, 10]
    b = batches[0]
    assert b.edge_index.shape == (2, 5 * A * (A - 1)) and b.edge_type.max() == 0
    assert np.all(b.T0 == 1000) and np.all(b.T1 == 300) and b.atoms.tolist() == list(range(A)) * 5
    np.testing.assert_allclose(b.x0.reshape(5, A, 3).mean(axis=1), 0, atol=1e-6)
    B2, A2, *_ = ti.thermo._molecule.split_batch(b, "atoms")
    assert (B2, A2) == (5, A)
    lat = d.LatentSamplerDataset(ref[0], T=800, n_samples=7, seed=3)
    lb = list(lat.batches(3))
    assert len(lb) == 2 and lb[0].T.dtype == np.int64 and np.all(lb[0].T == 800) and hasattr(lb[0], "atom_number")


def test_latent_to_ambient_chaining(tmp_path):
    """use_latent_trajs: end frames of the latent sampler's output become the ambient starting points (mdqm9_ambient.py:173-199)."""
    ti = pkg()
    d = ti.data
    rs = np.random.RandomState(4)
    A, n, n_step = 9, 11, 5
    samples = rs.standard_normal((n, n_step, A, 3)) + 3.0                 # not centred on purpose
    dl = rs.standard_normal(n).astype(np.float32)
    np.save(tmp_path / "samples_mol_00031_300k_forward.npy", samples)
    np.save(tmp_path / "dlogps_mol_00031_300k_forward.npy", dl)
    ds = d.MDQM9SamplerDataset("00031.npy", traj_path="unused", T0=300, T1=500, cutoff=1000, use_latent_trajs=True, n_latent_samples=8,
                               latent_traj_path=str(tmp_path))
    assert len(ds) == 8 and ds.data.shape == (8, A, 3)
    want = samples[:8, -1] - samples[:8, -1].mean(axis=1, keepdims=True)
    np.testing.assert_allclose(ds.data, want / d.SCALING_FACTOR, rtol=1e-6)
    np.testing.assert_allclose(ds.data0, samples[:8, 0] - samples[:8, 0].mean(axis=1, keepdims=True), rtol=1e-5, atol=1e-6)
    batch = next(ds.batches(4, shuffle=False))
    np.testing.assert_allclose(batch.latent_dlogp, dl[:4])
    assert batch.latent_z.shape == (4 * A, 3) and abs(batch.latent_z.reshape(4, A, 3).mean(axis=1)).max() < 1e-6
    assert np.all(batch.T0 == 300.0) and np.all(batch.T1 == 500.0)
    ds2 = d.MDQM9SamplerDataset("00031.npy", traj_path="unused", T0=300, scale=True, cutoff=1000, use_latent_trajs=True, n_latent_samples=3,
                                latent_traj_path=str(tmp_path))
    np.testing.assert_allclose(ds2.data, want[:3], rtol=1e-6)
    with pytest.raises(AssertionError):
        d.MDQM9SamplerDataset("00031.npy", traj_path="unused", use_latent_trajs=True)


def test_load_config_reads_the_reference_json_shape(tmp_path):
    ti = pkg()
    cfg = {"seed": 0, "n_features": 128, "score_layers": 5, "temp_length": 100, "batch_size": 12, "n_steps": 100, "atol": 1e-5,
           "rtol": 1e-5, "return_dlogp": 0, "T0s": [400, 500], "data_save_name": "x"}
    json.dump(cfg, open(tmp_path / "settings.json", "w"))
    ns = ti.drivers.load_config(str(tmp_path), "settings.json")
    assert ns.n_features == 128 and ns.atol == 1e-5 and ns.T0s == [400, 500]
    assert ti.drivers.load_config(str(tmp_path), "settings.json", ["--n_steps", "7"]).n_steps == 7


@pytest.mark.gpu
def test_drivers_end_to_end(tmp_path):
    """sample_ambient / sample_latent / sample_adw write the reference's files with the reference's shapes."""
    ti = pkg()
    d = ti.data
    g = load_golden("ambient_small")
    A, F, L = int(g["A"]), int(g["F"]), int(g["L"])
    traj = np.random.RandomState(1).standard_normal((8, 7, A, 3)) * 0.3
    os.makedirs(tmp_path / "test")
    np.save(tmp_path / "test" / "00031.npy", traj)
    ds = d.MDQM9SamplerDataset("00031.npy", str(tmp_path), "test", T0=1000, T1=300, scale=False, cutoff=1000)
    b = ti.thermo.ambient.cPaiNN(n_features=F, score_layers=L, temp_length=100)
    b.load_state_dict(ti.synthetic.painn_state_dict(0, F, L, 25, int(g["seed"])))
    cfg = types.SimpleNamespace(seed=0, batch_size=3, n_steps=6, atol=1e-5, rtol=1e-5, return_dlogp=0, method="euler",
                                data_save_path=str(tmp_path / "out"), data_save_name="t")
    samples, nfe = ti.drivers.sample_ambient(cfg, b, ds)
    assert samples.shape == (7, 6, A, 3) and nfe == 5
    assert np.load(tmp_path / "out" / "samples_t.npy").shape == (7, 6, A, 3)
    assert np.load(tmp_path / "out" / "latent_noises_t.npy").shape == (7, A, 3)
    assert np.load(tmp_path / "out" / "latent_dlogps_t.npy").shape == (7,)
    # every trajectory equals an individual rollout of that frame
    eng = b.engine_for(A, *[np.asarray(t) for t in ds.template], ds.atom_ids.astype(np.int32))
    order = np.random.RandomState(0).permutation(7)
    x0 = ds.data[order[:3]]
    path, _ = eng.rollout(x0, ti.synthetic.ambient_cond(3, A, 1000.0, (300.0,)), ti.engine.time_grid(0, 1, 6), scheme="euler")
    assert rel_l2(samples[:3], path.transpose(1, 0, 2, 3)) < 1e-6        # batches re-centre the (already centred) frames: 1-ulp inputs

    # return_dlogp=1: dlogps_{name}.npy holds the last row of the second state per trajectory (sample_ambient.py:96-100)
    cfg_d = types.SimpleNamespace(seed=0, batch_size=4, n_steps=3, atol=1e-5, rtol=1e-5, return_dlogp=1, method="euler",
                                  data_save_path=str(tmp_path / "outd"), data_save_name="d")
    samples_d, _ = ti.drivers.sample_ambient(cfg_d, b, ds)
    dl = np.load(tmp_path / "outd" / "dlogps_d.npy")
    assert samples_d.shape == (7, 3, A, 3) and dl.shape == (7,) and np.isfinite(dl).all() and np.abs(dl).max() > 0

    bl = ti.thermo.latent.cPaiNN(n_features=F, score_layers=L, temp_length=75)
    bl.precision = "f16x2"
    lds = d.LatentSamplerDataset(ds.data[0], T=800, n_samples=6, seed=2)
    cfg_l = types.SimpleNamespace(seed=0, batch_size=3, n_steps=4, atol=1e-5, rtol=1e-5, return_dlogp=0, data_save_path=str(tmp_path / "lat"),
                                  data_save_name="l")
    out = ti.drivers.sample_latent(cfg_l, bl, lds)
    assert out.shape == (6, 4, A, 3) and np.isfinite(out).all()
    assert np.load(tmp_path / "lat" / "samples_l_forward.npy").shape == (6, 4, A, 3)

    ga = load_golden("adw_ctor_h64")
    net = ti.thermo.adw.FCNetMultiBeta(1, 1, int(ga["hidden"]), int(ga["num_layers"]))
    net.load_state_dict({k[4:]: v for k, v in ga.items() if k.startswith("sd::")})
    cfg_a = types.SimpleNamespace(beta0s=[1.0], beta1s=[1.25], solver_type="euler", rtol=1e-4, atol=1e-4, n_step=len(ga["traj_grid"]),
                                  return_dlogp=1, data_save_path=str(tmp_path / "adw"), model_save_name="velocity", sampling_epoch=3)
    xs = ga["x"].astype(np.float32)[:, None]
    loader = [(xs[:8], np.ones((8, 1))), (xs[8:16], np.ones((8, 1)))]
    initial, samples = ti.drivers.sample_adw(cfg_a, net, loader)
    out_dir = tmp_path / "adw" / "velocity" / "beta_1.0_to_1.25"
    assert samples.shape == (len(ga["traj_grid"]), 16) and initial.shape == (16,)
    assert rel_l2(samples, ga["traj_euler"][:, :16]) < 1e-5
    assert np.load(out_dir / "dlogps_epoch_3.npy").shape == samples.shape
