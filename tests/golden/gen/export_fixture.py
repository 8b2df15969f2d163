"""Export the reference's only HDF5 fixture to a small .npz (data, not code).

Run ONCE in the build container with the interpreter that has h5py:
    /opt/conda/bin/python3.9 tests/golden/gen/export_fixture.py
Source: /root/reference/tests/hdf5/1ATN_residue.hdf5 (the fixture used by the
reference's tests/test_nn.py:38).  Output: tests/golden/fixture_1ATN.npz with
keys "<mol>/<dataset path>" for exactly the datasets HDF5DataSet.load_one_graph
(DataSet.py:231-366) reads, plus every score and every node feature.
Byte-string datasets (residue names, edge types) are skipped: the hot path never
reads them.
"""
import sys
import numpy as np
import h5py

SRC = "/root/reference/tests/hdf5/1ATN_residue.hdf5"
DST = "tests/golden/fixture_1ATN.npz"

out = {}
with h5py.File(SRC, "r") as f:
    mols = list(f.keys())
    out["__mols__"] = np.array(mols)
    for mol in mols:
        def visit(name, obj, mol=mol):
            if isinstance(obj, h5py.Dataset) and obj.dtype.kind in "fiub":
                out[f"{mol}/{name}"] = obj[()]
        f[mol].visititems(visit)
np.savez_compressed(DST, **out)
print("wrote", DST, len(out), "arrays")
