"""Host-side data plumbing next to the hot path: per-species edge templates, batches, datasets.

The reference builds its graphs with PyG / torch_cluster / RDKit, none of which exist offline; the sampler itself only
needs ONE molecule's template (every batch is one species).  Everything here is numpy.

Restated from (paths relative to /root/reference):
  mdqm9/thermo/utils.py:69-125        AddRadiusGraph (edge type 0), AddBondGraph (edge type = bond order), Coalesce(reduce="max")
  mdqm9/data/mdqm9_ambient.py:14-16   scaling constants;  :173-211 trajectory loading, COM removal, scaling
  mdqm9/data/mdqm9_ambient.py:228-249 bond list from the SDF record (both directions, bond order cast to long)
  mdqm9/data/mdqm9_ambient.py:148-170 / mdqm9_latent.py:163-205   per-item processing (COM removal, T0/T1/T features)
Parity status: unpinned -- no trajectory or SDF files exist offline; tests use synthetic files of the documented layout.
"""
from __future__ import annotations

import os
import types

import numpy as np

SCALING_FACTOR = 0.20754094                  # mdqm9_ambient.py:14  (general MDQM9)
SCALING_FACTOR_31 = 0.09729941375            # :15  molecule 00031
SCALING_FACTOR_10506 = 0.13163184188306332   # :16  molecule 10506
TEMPERATURES = tuple(range(300, 1001, 100))  # index of the leading axis of the trajectory files (mdqm9_ambient.py:131)


# ------------------------------------------------------------------------------------------------- graph template
def radius_edges(x: np.ndarray, cutoff: float):
    """All ordered pairs (j -> i), i != j, with |x_i - x_j| <= cutoff (torch_geometric.nn.radius_graph without self loops;
    the shipped configs use cutoff = 1000, i.e. fully connected)."""
    x = np.asarray(x, np.float64)
    d = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=-1)
    src, dst = np.nonzero((d <= cutoff) & ~np.eye(len(x), dtype=bool))
    return src.astype(np.int64), dst.astype(np.int64)


def coalesce_max(src, dst, etype):
    """torch_geometric.utils.coalesce(edge_index, edge_type, reduce="max"): sort by (src, dst), merge duplicates with max."""
    src, dst, etype = (np.asarray(a, np.int64) for a in (src, dst, etype))
    n = int(max(src.max(initial=-1), dst.max(initial=-1))) + 1
    key = src * max(n, 1) + dst
    order = np.argsort(key, kind="stable")
    key, src, dst, etype = key[order], src[order], dst[order], etype[order]
    first = np.concatenate([[True], key[1:] != key[:-1]]) if key.size else np.zeros(0, bool)
    group = np.cumsum(first) - 1
    out_type = np.zeros(int(first.sum()), np.int64)
    np.maximum.at(out_type, group, etype)
    return src[first].astype(np.int32), dst[first].astype(np.int32), out_type.astype(np.int32)


def build_edge_template(x: np.ndarray, cutoff: float = np.inf, bond_index=None, bonds=None):
    """(edge_src, edge_dst, edge_type) of ONE molecule in the reference order: radius graph (type 0) + bond graph (type =
    bond order), coalesced with max, sorted by (src, dst) -- what `add_radius_graph`, `add_bond_graph`, `coalesce` produce
    (mdqm9_ambient.py:165-169).  `bond_index` [2, 2*Nb] and `bonds` [2*Nb] as returned by `bonds_from_molblock`."""
    rs, rd = radius_edges(x, cutoff)
    src, dst, et = rs, rd, np.zeros(rs.size, np.int64)
    if bond_index is not None:
        bi = np.asarray(bond_index, np.int64)
        src, dst = np.concatenate([src, bi[0]]), np.concatenate([dst, bi[1]])
        et = np.concatenate([et, np.asarray(bonds, np.int64)])
    if et.size and (et.min() < 0 or et.max() > 3):
        raise ValueError("edge types must be in 0..3 (the edge-type embedding has 4 rows, cpainn.py:70)")
    return coalesce_max(src, dst, et)


def read_sdf_record(path: str, index: int) -> str:
    """Text of record `index` of a multi-molecule SDF file ('$$$$' separated), like ``Chem.SDMolSupplier(...)[index]``."""
    with open(path) as f:
        records = f.read().split("$$$$\n")
    return records[index]


def bonds_from_molblock(text: str):
    """(bond_index [2, 2*Nb], bonds [2*Nb]) from a V2000 mol block: both directions, bond order as integer
    (get_bond_index_and_bonds, mdqm9_ambient.py:228-249).  RDKit's aromaticity perception is NOT reproduced: a file that marks
    aromatic bonds with type 4 maps them to 1 (the reference casts GetBondTypeAsDouble() = 1.5 to long), a Kekule file keeps
    its alternating 1/2 orders."""
    lines = text.splitlines()
    counts = lines[3]
    n_atoms, n_bonds = int(counts[0:3]), int(counts[3:6])
    begin, end, order = [], [], []
    for ln in lines[4 + n_atoms: 4 + n_atoms + n_bonds]:
        a, b, t = int(ln[0:3]) - 1, int(ln[3:6]) - 1, int(ln[6:9])
        begin.append(a); end.append(b); order.append(1 if t == 4 else t)
    begin, end, order = (np.asarray(v, np.int64) for v in (begin, end, order))
    return np.stack([np.concatenate([begin, end]), np.concatenate([end, begin])]), np.concatenate([order, order])


# ------------------------------------------------------------------------------------------------- batches
def _remove_com(x):
    return x - x.mean(axis=-2, keepdims=True)


def make_batch(variant: str, x0, template, *, T0=None, T1=None, T=None, atom_ids=None, latent_z=None, latent_dlogp=None):
    """Attribute bag with the fields the reference batches carry (numpy arrays): x0/x [N,3] (COM removed per molecule,
    mdqm9_ambient.py:161), atoms | atom_number [N], T0/T1 | T [N], edge_index [2,E], edge_type [E], batch [N]."""
    x0 = _remove_com(np.asarray(x0, np.float32))
    B, A, _ = x0.shape
    src, dst, et = template
    ids = np.arange(A, dtype=np.int64) if atom_ids is None else np.asarray(atom_ids, np.int64)
    off = (np.arange(B, dtype=np.int64) * A)[:, None]
    b = types.SimpleNamespace()
    b.x0 = x0.reshape(B * A, 3)
    b.x = b.x0.copy()
    b.edge_index = np.stack([(src[None, :] + off).ravel(), (dst[None, :] + off).ravel()]).astype(np.int64)
    b.edge_type = np.tile(np.asarray(et, np.int64), B)
    b.batch = np.repeat(np.arange(B, dtype=np.int64), A)
    if variant == "ambient":
        b.atoms = np.tile(ids, B)
        b.T0 = np.repeat(np.broadcast_to(np.asarray(T0, np.float32), (B,)), A)
        b.T1 = np.repeat(np.broadcast_to(np.asarray(T1, np.float32), (B,)), A)
        b.latent_z = _remove_com(np.zeros_like(x0) if latent_z is None else np.asarray(latent_z, np.float32)).reshape(B * A, 3)
        b.latent_dlogp = np.zeros(B, np.float32) if latent_dlogp is None else np.asarray(latent_dlogp, np.float32)
    elif variant == "latent":
        b.atom_number = np.tile(ids, B)
        if T is not None:
            b.T = np.repeat(np.broadcast_to(np.asarray(T, np.int64), (B,)), A)      # int64 like mdqm9_latent.py:184
    else:
        raise ValueError("variant must be 'ambient' or 'latent'")
    return b


def load_trajectory(traj_path: str, split: str, traj_filename: str, T: int, scale: bool) -> np.ndarray:
    """[n_frames, A, 3] frames at temperature T from ``{traj_path}/{split}/{traj_filename}`` ([8, n_frames, A, 3], axis 0 =
    TEMPERATURES), centred, optionally scaled (get_mdqm9_trajs, mdqm9_ambient.py:202-211)."""
    trajs = np.load(os.path.join(traj_path, split, traj_filename), mmap_mode="r")[TEMPERATURES.index(int(T))]
    trajs = _remove_com(np.asarray(trajs, np.float64))
    if scale:
        trajs = trajs * (SCALING_FACTOR_31 if traj_filename == "00031.npy" else SCALING_FACTOR_10506)
    return trajs.astype(np.float32)


def load_latent_trajectory(latent_traj_path: str, traj_filename: str, T: int, n_samples: int, scale: bool):
    """(z, x, dlogp0): first and last frame of the latent sampler's output ``samples_mol_{id}_{T}k_forward.npy``
    [n, n_step, A, 3] and, for molecule 00031, ``dlogps_mol_{id}_{T}k_forward.npy`` (zeros otherwise); both centred, the end frames
    divided by SCALING_FACTOR unless `scale` (get_latent_mdqm9_trajs, mdqm9_ambient.py:173-199)."""
    assert traj_filename in {"00031.npy", "10506.npy"}
    idx = traj_filename[:-4]
    samples = np.load(os.path.join(latent_traj_path, f"samples_mol_{idx}_{T}k_forward.npy"), mmap_mode="r")
    z = _remove_com(np.asarray(samples[:n_samples, 0], np.float64))
    x = _remove_com(np.asarray(samples[:n_samples, -1], np.float64))
    if traj_filename == "00031.npy":
        dlogp0 = np.asarray(np.load(os.path.join(latent_traj_path, f"dlogps_mol_{idx}_{T}k_forward.npy"))[:n_samples], np.float32)
    else:
        dlogp0 = np.zeros(len(x), np.float32)
    if not scale:
        x = x / SCALING_FACTOR
    return z.astype(np.float32), x.astype(np.float32), dlogp0


class MDQM9SamplerDataset:
    """Ambient sampling dataset (MDQM9SamplerDataset, mdqm9_ambient.py:110-170): frames at T0 as starting points, to be carried to
    T1.  Bonds come from a mol block (`sdf_path/sdf_filename`, record = molecule id) or are passed explicitly.  With
    `use_latent_trajs` the starting points are the end states of the latent sampler (latent -> ambient chaining): its first frames
    and dlogps travel along as `latent_z` / `latent_dlogp`."""

    def __init__(self, traj_filename, traj_path, split="test", T0=300, T1=400, scale=False, cutoff=np.inf, sdf_path=None,
                 sdf_filename="mdqm9.sdf", bond_index=None, bonds=None, use_latent_trajs=False, n_latent_samples=10_000,
                 latent_traj_path=""):
        assert split in {"train", "val", "test"}
        if use_latent_trajs:
            assert latent_traj_path != "", "latent_traj_path must be provided if use_latent_trajs is True"
            self.data0, self.data, self.dlogp0 = load_latent_trajectory(latent_traj_path, traj_filename, T0, n_latent_samples, scale)
        else:
            self.data = load_trajectory(traj_path, split, traj_filename, T0, scale)
            self.data0, self.dlogp0 = np.zeros_like(self.data), np.zeros(len(self.data), np.float32)      # dummies, like the reference
        self.T0, self.T1 = float(T0), float(T1)
        if bond_index is None and sdf_path is not None:
            bond_index, bonds = bonds_from_molblock(read_sdf_record(os.path.join(sdf_path, sdf_filename), int(traj_filename.split(".")[0])))
        self.template = build_edge_template(self.data[0], cutoff, bond_index, bonds)
        self.atom_ids = np.arange(self.data.shape[1])                        # distinguish=True (mdqm9_ambient.py:219-220)

    def __len__(self):
        return len(self.data)

    def batches(self, batch_size, shuffle=True, seed=0, drop_last=False):
        order = np.random.RandomState(seed).permutation(len(self)) if shuffle else np.arange(len(self))
        for i in range(0, len(order), batch_size):
            idx = order[i:i + batch_size]
            if drop_last and len(idx) < batch_size:
                return
            yield make_batch("ambient", self.data[idx], self.template, T0=self.T0, T1=self.T1, atom_ids=self.atom_ids,
                             latent_z=self.data0[idx], latent_dlogp=self.dlogp0[idx])


class LatentSamplerDataset:
    """Latent sampling dataset (SamplerDataset, mdqm9_latent.py:116-205): Gaussian noise x0 (COM removed) -> Boltzmann at T."""

    def __init__(self, x1, T=300, n_samples=1000, cutoff=np.inf, bond_index=None, bonds=None, seed=0):
        self.x1 = np.asarray(x1, np.float32)                                   # one reference frame: fixes A and the graph
        self.T, self.n, self.seed = int(T), int(n_samples), int(seed)
        self.template = build_edge_template(self.x1, cutoff, bond_index, bonds)
        self.atom_ids = np.arange(self.x1.shape[0])

    def __len__(self):
        return self.n

    def batches(self, batch_size, seed=None, drop_last=True):
        rs = np.random.RandomState(self.seed if seed is None else seed)
        for i in range(0, self.n, batch_size):
            nb = min(batch_size, self.n - i)
            if drop_last and nb < batch_size:
                return
            yield make_batch("latent", rs.standard_normal((nb,) + self.x1.shape), self.template, T=self.T, atom_ids=self.atom_ids)
