"""The pair-major edge template (csrc/pair_template.hpp), checked on the CPU: the builder is compiled with g++ into a small
harness and its output is held to the invariants the kernel (csrc/painn_pair_kernel.hpp) relies on."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pair") / "pair_template_dump")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "harness", "pair_template_dump.cpp")])

    def run(A, src, dst, et):
        text = f"{A} {len(src)}\n" + "".join(f"{s} {d} {t}\n" for s, d, t in zip(src, dst, et))
        out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
        if out[0].strip() == "none":
            return None
        _, G, nblk, kmax = out[0].split()
        G, nblk, kmax = int(G), int(nblk), int(kmax)
        rows = np.array(out[1].split(), dtype=np.int64).reshape(nblk, 16)
        slots = np.array(out[2].split(), dtype=np.int64).reshape(nblk, 16)
        pos = np.array(out[3].split(), dtype=np.int64).reshape(G, A, A)
        plist = np.array(out[4].split(), dtype=np.int64).reshape(G, A, kmax)
        return G, nblk, rows, slots, pos, plist
    return run


def check_template(A, src, dst, et, tpl):
    G, nblk, rows, slots, pos, plist = tpl
    etype = {(s, d): t for s, d, t in zip(src, dst, et)}
    seen, owner = set(), {}
    for b in range(nblk):
        I, J = slots[b, :4], slots[b, 4:8]
        assert np.all(slots[b, 8:] == -1)
        keys = lambda s: [(int(w) >> 8 & 0x3FFFFF, int(w) & 255) for w in s if w >= 0]
        assert len(set(keys(I))) == len(keys(I)) and len(set(keys(J))) == len(keys(J))      # an atom at most once per side
        for k in range(8):                                   # partial row (block, slot) -> the atom whose sum it holds
            if slots[b, k] >= 0:
                owner[8 * b + k] = (int(slots[b, k]) >> 8 & 0x3FFFFF, int(slots[b, k]) & 255)
        for a in range(4):
            for c in range(4):
                w = int(rows[b, 4 * a + c])
                mI, aI, mJ, aJ, ty = w >> 1 & 7, w >> 4 & 31, w >> 9 & 7, w >> 12 & 31, w >> 17 & 3
                assert mI < G and mJ < G and aI < A and aJ < A                                   # loads of invalid rows stay inside the group
                if I[a] >= 0:
                    assert (int(I[a]) >> 8 & 0x3FFFFF, int(I[a]) & 255) == (mI, aI)
                if J[c] >= 0:
                    assert (int(J[c]) >> 8 & 0x3FFFFF, int(J[c]) & 255) == (mJ, aJ)
                if w & 1:
                    assert I[a] >= 0 and J[c] >= 0 and mI == mJ and (aI, aJ) in etype and etype[(aI, aJ)] == ty
                    for e in ((mI, aI, aJ), (mI, aJ, aI)):
                        assert e not in seen
                        seen.add(e)
                    assert pos[mI, aI, aJ] == (2 * b) * 16 + 4 * a + c and pos[mI, aJ, aI] == (2 * b + 1) * 16 + 4 * a + c
    assert seen == {(m, s, d) for m in range(G) for s, d in zip(src, dst)}            # every directed edge of every molecule exactly once
    # the partial lists: every occupied (block, slot) appears in exactly one list, its atom's, in walk order
    listed = {}
    for m in range(G):
        for a in range(A):
            ids = [int(v) for v in plist[m, a] if v >= 0]
            assert ids == sorted(ids) and list(plist[m, a][:len(ids)]) == ids              # -1 only as padding at the end
            for i in ids:
                assert i not in listed
                listed[i] = (m, a)
    assert listed == owner
    return sum(int(w) & 1 for w in rows.ravel()) / (nblk * 16.0)


def test_complete_graphs_meet_the_slot_bound(dump):
    syn = pkg().synthetic
    expect = {18: (4, 45), 9: (1, 3), 25: (1, 21)}            # A: (G, blocks per group) -- 11.25 / 3 / 21 blocks per molecule
    for A in (2, 3, 5, 9, 12, 17, 18, 25, 32):
        src, dst, et = syn.fully_connected_template(A)
        tpl = dump(A, src, dst, et)
        assert tpl is not None
        fill = check_template(A, list(src), list(dst), list(et), tpl)
        G, nblk = tpl[0], tpl[1]
        per_atom = -(-(A - 1) // 4)                          # slot places an atom needs at least
        assert nblk / G >= A * per_atom / 8.0 - 1e-9
        if A in expect:
            assert (G, nblk) == expect[A], (A, G, nblk, fill)


def test_sparse_and_asymmetric_graphs(dump):
    rs = np.random.RandomState(3)
    for A, p in ((9, 0.5), (18, 0.3), (25, 0.6), (18, 0.9)):
        adj = np.triu(rs.rand(A, A) < p, 1)
        for a in range(A - 1):                                # a chain keeps every atom connected
            adj[a, a + 1] = True
        i, j = np.nonzero(adj)
        ty = rs.randint(0, 4, i.size)
        src, dst, et = np.concatenate([i, j]), np.concatenate([j, i]), np.concatenate([ty, ty])
        order = np.lexsort((dst, src))
        src, dst, et = src[order], dst[order], et[order]
        tpl = dump(A, src, dst, et)
        assert tpl is not None
        check_template(A, [int(v) for v in src], [int(v) for v in dst], [int(v) for v in et], tpl)
    # not the symmetric closure of an undirected graph -> no pair template (the directed kernels serve it)
    assert dump(3, [0, 1, 1], [1, 0, 2], [0, 0, 0]) is None                     # 2 -> 1 missing (odd edge count)
    assert dump(3, [0, 1, 1, 2], [1, 0, 2, 1], [0, 1, 0, 0]) is None            # types differ between the two directions
    assert dump(3, [0, 1, 0, 2], [1, 0, 2, 2], [0, 0, 0, 0]) is None            # self loop
