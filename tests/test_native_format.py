"""The native on-disk formats (container.py): the lossless tree mirror of the reference's HDF5 schema read by
GraphStore without h5py, multi-file datasets, and (emulated kernels) the resident-set image with its cached topology."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from helpers import GOLDEN, NODE_FEATURES, ROOT

DRGS = os.path.join(GOLDEN, "1ATN_residue.drgs")
NPZ = os.path.join(GOLDEN, "fixture_1ATN.npz")
REF_H5 = "/root/reference/tests/hdf5/1ATN_residue.hdf5"
CONDA = "/opt/conda/bin/python3.9"


def test_container_roundtrip(tmp_path):
    from deeprank_gnn_amd.container import read_container, read_header, write_container
    rng = np.random.default_rng(0)
    sec = {"a/x": rng.normal(size=(7, 3)).astype(np.float32), "a/i": np.arange(11, dtype=np.int64),
           "b/names": np.array([[b"A", b"12", b"GLY"], [b"B", b"7", b"TRP"]]), "b/scalar": np.float64(3.5),
           "b/empty": np.zeros((0, 2), dtype=np.int32), "b/flag": np.array(True)}
    path = str(tmp_path / "t.drgs")
    write_container(path, sec, meta={"hello": [1, 2, 3]})
    hdr = read_header(path)
    assert all(d["offset"] % 64 == 0 for d in hdr["sections"].values())
    meta, got = read_container(path)
    assert meta == {"hello": [1, 2, 3]}
    for k, v in sec.items():
        v = np.asarray(v)
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    _, only_b = read_container(path, prefix="b/", mmap=True)
    assert sorted(only_b) == ["b/empty", "b/flag", "b/names", "b/scalar"]


def test_tree_mirror_matches_the_npz_fixture_and_keeps_strings(tmp_path):
    """tests/golden/1ATN_residue.drgs was converted from the reference's tests/hdf5/1ATN_residue.hdf5 by
    tools/hdf5_to_native.py; fixture_1ATN.npz is the numeric export of the same file (round 1)."""
    from deeprank_gnn_amd.dataset import GraphStore
    a, b = GraphStore(DRGS), GraphStore(NPZ)
    assert a.mols() == b.mols() and len(a.mols()) == 10
    for mol in b.mols():
        for key, val in b._mols[mol].items():
            got = a.get(mol, key)
            assert got.dtype == val.dtype and np.array_equal(got, val), (mol, key)
        assert a.get(mol, "nodes").dtype.kind == "S" and a.get(mol, "nodes").shape[1] == 3      # chain, resSeq, resName
        assert a.get(mol, "edges").shape[1:] == (2, 3)
    out = str(tmp_path / "again.drgs")
    a.save_native(out)
    c = GraphStore(out)
    for mol in a.mols():
        assert sorted(c._mols[mol]) == sorted(a._mols[mol])
        for key, val in a._mols[mol].items():
            assert np.array_equal(c.get(mol, key), val) and c.get(mol, key).dtype == val.dtype


def test_dataset_from_native_equals_dataset_from_npz_and_takes_a_list():
    from deeprank_gnn_amd.dataset import GraphDataSet
    kw = dict(node_feature=NODE_FEATURES, edge_feature=["dist"], target="irmsd")
    one, ref = GraphDataSet(DRGS, **kw), GraphDataSet(NPZ, **kw)
    assert len(one) == len(ref) == 10
    for i in (0, 4, 9):
        g, h = one[i], ref[i]
        for k in ("x", "edge_index", "edge_attr", "y", "pos", "cluster0", "cluster1", "internal_edge_index"):
            assert torch.equal(getattr(g, k), getattr(h, k)), k
    # several files (reference DataSet.py:116-118), `index` applied per file (DataSet.py:388-398)
    two = GraphDataSet([DRGS, NPZ], index=[0, 3, 5], **kw)
    assert len(two) == 6 and two.mols == [ref.mols[i] for i in (0, 3, 5)] * 2
    assert torch.equal(two[4].x, ref[3].x) and two[4].mol == ref.mols[3]
    with pytest.raises(ValueError):
        GraphDataSet([], **kw)


@pytest.mark.skipif(not (os.path.exists(REF_H5) and os.path.exists(CONDA)), reason="needs the build container (h5py)")
def test_hdf5_converters_are_lossless(tmp_path):
    """reference .hdf5 -> native -> .hdf5: every dataset identical (names, dtypes, shapes, values); the committed
    fixture is what the converter writes."""
    nat, back = str(tmp_path / "a.drgs"), str(tmp_path / "b.hdf5")
    subprocess.check_call([CONDA, os.path.join(ROOT, "tools", "hdf5_to_native.py"), REF_H5, nat])
    subprocess.check_call([CONDA, os.path.join(ROOT, "tools", "native_to_hdf5.py"), nat, back])
    check = ("import h5py, numpy as np, sys\n"
             "a, b = h5py.File(sys.argv[1], 'r'), h5py.File(sys.argv[2], 'r')\n"
             "na, nb = [], []\n"
             "a.visititems(lambda n, o: na.append(n) if isinstance(o, h5py.Dataset) else None)\n"
             "b.visititems(lambda n, o: nb.append(n) if isinstance(o, h5py.Dataset) else None)\n"
             "assert sorted(na) == sorted(nb) and len(na) > 200\n"
             "for n in na:\n"
             "    x, y = np.asarray(a[n][()]), np.asarray(b[n][()])\n"
             "    assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), n\n")
    subprocess.check_call([CONDA, "-c", check, REF_H5, back])
    from deeprank_gnn_amd.container import read_container
    _, fresh = read_container(nat)
    _, kept = read_container(DRGS)
    assert sorted(fresh) == sorted(kept)
    for k in fresh:
        assert np.array_equal(fresh[k], kept[k]), k


def test_resident_set_image_with_cached_topology(tmp_path):
    """save_native / load_native of the uploaded image + its cached topology (kernels emulated on the CPU): the
    reloaded set trains bit-identically out of the stored topology, without building anything."""
    from emu_api import emu
    from collate_check import ragged_graphs
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.trainer import FusedTrainer
    import copy
    api = emu()
    graphs = ragged_graphs(9, 12)
    rs = ResidentGraphSet(graphs, "cpu", api=api)
    path = str(tmp_path / "set.drgs")
    rs.save_native(path)
    from deeprank_gnn_amd.container import read_header
    hdr = read_header(path)
    assert {"set/x", "set/edge_index", "set/node_ptr", "topo/ws_i32", "topo/ws_f32"} <= set(hdr["sections"])
    assert hdr["meta"]["topology"]["arrays_i32"][:2] == ["NPTR", "EPTR"]
    back = ResidentGraphSet.load_native(path, "cpu", api=api)
    assert back.mols == rs.mols and torch.equal(back.x, rs.x) and torch.equal(back.edge_index, rs.edge_index)
    assert True in back._topo_cache                          # adopted from the file, not rebuilt
    assert torch.equal(back._topo_cache[True].topo.ws_i32, rs.topology_cache(True).topo.ws_i32)
    torch.manual_seed(0)
    net = sGAT(12, 1, 1)
    ta, tb = FusedTrainer(net, lr=0.01, api=api), FusedTrainer(copy.deepcopy(net), lr=0.01, api=api)
    order = [3, 1, 8, 0, 5, 2, 7, 4, 6]
    la, pa = ta.train_epoch(rs, order, 4)                    # rebuilt per mini-batch
    lb, pb = tb.train_epoch(back, order, 4, cached=True)     # out of the stored topology
    assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(ta.flat_p, tb.flat_p)
