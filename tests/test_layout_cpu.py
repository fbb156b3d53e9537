"""CPU tier: the edge-tile layout (hd_topology_layout = the host half of hd_topology_create).

SURVEY.md section 8e asks for results independent of world size.  The kernels' per-node sums depend on (a) which edge
rows are added into one partial sum ("part"), in which order, (b) the order the parts are added, (c) the position of
a row inside its 32-row tile modulo 4 (k_edge.hpp).  These tests pin that all three are functions of the molecule
alone - whatever batch, shard or rank it is laid out in - and that every unmasked edge appears exactly once.
"""
import ctypes as C

import numpy as np
import pytest

from hierdiff_amd import _lib


@pytest.fixture(scope="module")
def lib():
    from hierdiff_amd import build
    build.build(verbose=False)
    return _lib.load()


def layout(lib, node_mask, edge_mask=None):
    nm = np.ascontiguousarray(node_mask, dtype=np.uint8)
    B, N = nm.shape
    em = None if edge_mask is None else np.ascontiguousarray(edge_mask, dtype=np.uint8)
    counts = (C.c_longlong * 5)()
    emp = None if em is None else em.ctypes.data
    _lib.check(lib.hd_topology_layout(nm.ctypes.data, emp, B, N, counts, None, None, None, None, None, None))
    M, E, tiles, parts, rows = (int(v) for v in counts)
    ei, ej, sp = (np.zeros(rows, np.int32) for _ in range(3))
    eseg = np.zeros(rows, np.uint8)
    nseg = np.zeros(tiles, np.int32)
    pstart = np.zeros(M + 1, np.int32)
    _lib.check(lib.hd_topology_layout(nm.ctypes.data, emp, B, N, counts, ei.ctypes.data, ej.ctypes.data, eseg.ctypes.data,
                                      sp.ctypes.data, nseg.ctypes.data, pstart.ctypes.data))
    return dict(M=M, E=E, tiles=tiles, parts=parts, rows=rows, ei=ei, ej=ej, eseg=eseg, seg_part=sp, nseg=nseg,
                pstart=pstart, B=B, N=N, node_mask=nm, edge_mask=em)


def signatures(L):
    """Per molecule: for every node (molecule-relative id) the ordered list of its parts, each part =
    (row offset in tile mod 4, ordered molecule-relative sender ids)."""
    nm = L["node_mask"]
    B, N = nm.shape
    # compact id -> (molecule, local id): active nodes are numbered in flat order
    if L["edge_mask"] is None:
        active = nm.astype(bool)
    else:
        em = L["edge_mask"].reshape(B, N, N).astype(bool)
        active = nm.astype(bool) | em.any(2) | em.any(1)
    flat = np.flatnonzero(active.reshape(-1))
    assert len(flat) == L["M"]
    mol_of, loc_of = flat // N, flat % N
    part_rows = {}
    for t in range(L["tiles"]):
        for r in range(32):
            s = int(L["eseg"][t * 32 + r])
            if s == 255:
                continue
            assert s < L["nseg"][t]
            pid = int(L["seg_part"][t * 32 + s])
            part_rows.setdefault(pid, []).append((t, r))
    sig = [dict() for _ in range(B)]
    seen_edges = set()
    for node in range(L["M"]):
        parts = []
        for pid in range(L["pstart"][node], L["pstart"][node + 1]):
            rows = part_rows.pop(pid)
            t0, r0 = rows[0]
            assert all(t == t0 for t, _ in rows) and [r for _, r in rows] == list(range(r0, r0 + len(rows)))
            senders = []
            for t, r in rows:
                assert L["ei"][t * 32 + r] == node
                j = int(L["ej"][t * 32 + r])
                assert mol_of[j] == mol_of[node]
                senders.append(int(loc_of[j]))
                e = (node, j)
                assert e not in seen_edges
                seen_edges.add(e)
            parts.append((r0 % 4, tuple(senders)))
        sig[mol_of[node]][int(loc_of[node])] = tuple(parts)
    assert not part_rows, "parts not owned by any node"
    assert len(seen_edges) == L["E"]
    return sig, seen_edges, (mol_of, loc_of)


def canonical_edges(nm):
    B, N = nm.shape
    return {(b, i, j) for b in range(B) for i in range(N) for j in range(N) if nm[b, i] and nm[b, j] and i != j}


def mask_of(n_list, N=None):
    N = N or max(n_list)
    return (np.arange(N)[None, :] < np.asarray(n_list)[:, None]).astype(np.uint8)


@pytest.mark.parametrize("n_list", [[30] * 8, [1], [2, 2, 2, 1, 2], [33, 40, 5], [83, 3], [17, 9, 30, 12, 25, 7, 14, 21],
                                    [6, 6, 6, 6, 6, 6, 6], [3, 4, 3, 1, 2, 5, 30, 2, 3]])
def test_every_edge_once_and_parts_in_order(lib, n_list):
    nm = mask_of(n_list)
    L = layout(lib, nm)
    sig, seen, (mol_of, loc_of) = signatures(L)
    got = {(int(mol_of[i]), int(loc_of[i]), int(loc_of[j])) for i, j in seen}
    assert got == canonical_edges(nm)
    assert L["E"] == sum(n * (n - 1) for n in n_list)
    assert L["tiles"] % 4 == 0 or L["E"] == 0                # whole workgroups of four 32-row tiles
    # a node's senders appear in ascending order across its parts (the reference's row-major edge order)
    for b, n in enumerate(n_list):
        for i in range(n):
            order = [j for _, snd in sig[b][i] for j in snd]
            assert order == [j for j in range(n) if j != i]


def test_layout_of_a_molecule_is_independent_of_the_batch(lib):
    rng = np.random.default_rng(7)
    n_all = [int(v) for v in rng.integers(1, 49, size=40)] + [30, 30, 83, 2, 1]
    N = max(n_all)
    full, _, _ = signatures(layout(lib, mask_of(n_all, N)))
    # every contiguous shard (what sharding.py hands a rank), with a different padded width, reproduces the signatures
    for lo, hi in [(0, 20), (20, 45), (7, 8), (44, 45), (13, 31)]:
        sub = n_all[lo:hi]
        shard, _, _ = signatures(layout(lib, mask_of(sub, max(sub) + 3)))
        for k in range(hi - lo):
            assert shard[k] == full[lo + k], (lo, k)
    # ... and so does a permuted batch
    perm = rng.permutation(len(n_all))
    shuf, _, _ = signatures(layout(lib, mask_of([n_all[p] for p in perm], N)))
    for k, p in enumerate(perm):
        assert shuf[k] == full[p]


def test_padding_cost_of_alignment(lib):
    """Molecule-aligned cuts cost padding rows only in shared tail tiles: < 1 % at the headline shape."""
    L = layout(lib, mask_of([30] * 256))
    assert L["E"] == 256 * 870
    assert L["rows"] <= 1.01 * L["E"] + 128
    # GEOM-like sizes (mean ~15): tails of 8 molecules share a tile
    rng = np.random.default_rng(2022)
    n_list = [int(v) for v in np.clip(rng.poisson(15, 256), 1, 48)]
    L = layout(lib, mask_of(n_list, 48))
    assert L["rows"] <= 1.08 * L["E"] + 128


def test_general_edge_mask_layout(lib):
    """Block-diagonal mask with a self edge, an asymmetric hole and an edge-only (node-masked-out) participant."""
    nm = mask_of([9, 6, 12], 12)
    B, N = nm.shape
    em = np.zeros((B, N, N), np.uint8)
    for b, n in enumerate([9, 6, 12]):
        em[b, :n, :n] = 1 - np.eye(n, dtype=np.uint8)
    em[0, :4, 4:9] = 0; em[0, 4:9, :4] = 0
    em[1, 2, 2] = 1
    em[2, 0, 1] = 0
    em[1, 7, 0] = 1                      # node 7 of molecule 1 is masked out but receives an edge: becomes active
    L = layout(lib, nm, em)
    sig, seen, (mol_of, loc_of) = signatures(L)
    got = {(int(mol_of[i]), int(loc_of[i]), int(loc_of[j])) for i, j in seen}
    want = {(b, i, j) for b in range(B) for i in range(N) for j in range(N) if em[b, i, j]}
    assert got == want and L["E"] == int(em.sum())
    assert L["M"] == int(nm.sum()) + 1


def test_empty_and_bad_arguments(lib):
    L = layout(lib, mask_of([1, 1, 1]))
    assert L["E"] == 0 and L["parts"] == 0 and L["tiles"] == 1 and np.all(L["eseg"] == 255)
    counts = (C.c_longlong * 5)()
    nm = np.ones((1, 2), np.uint8)
    assert lib.hd_topology_layout(nm.ctypes.data, None, 0, 2, counts, None, None, None, None, None, None) == -1
    assert lib.hd_topology_layout(None, None, 1, 2, counts, None, None, None, None, None, None) == -1
