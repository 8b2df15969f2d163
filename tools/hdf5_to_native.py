#!/usr/bin/env python
"""HDF5 graph file (the reference's schema, written by Graph.nx2h5, reference Graph.py:61-139) -> native .drgs container.

Run it where h5py exists (build container: /opt/conda/bin/python3.9 tools/hdf5_to_native.py in.hdf5 out.drgs); the
result is read by deeprank_gnn_amd.dataset.GraphStore WITHOUT h5py (the MI355X image has none).  Lossless: every dataset
of every molecule group keeps its path, dtype and shape, byte-string datasets (nodes, edges) included.
Needs numpy + h5py only: container.py is loaded by file path, the package (which imports torch) is not."""
import importlib.util
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("drgs_container", os.path.join(HERE, "..", "deeprank-gnn_amd", "container.py"))
container = importlib.util.module_from_spec(spec)
spec.loader.exec_module(container)


def convert(src, dst):
    sections, mols = {}, []
    with h5py.File(src, "r") as f:
        for mol in f.keys():
            mols.append(mol)

            def visit(name, obj, mol=mol):
                if isinstance(obj, h5py.Dataset):
                    sections["tree/%s/%s" % (mol, name)] = np.asarray(obj[()])
            f[mol].visititems(visit)
    container.write_container(dst, sections, meta={"kind": "tree", "mols": mols, "source": os.path.basename(src),
                                                   "schema": "deeprank_gnn Graph.nx2h5 (reference Graph.py:61-139)"})
    return len(mols), len(sections)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    n_mol, n_ds = convert(sys.argv[1], sys.argv[2])
    print("%s: %d molecules, %d datasets -> %s" % (sys.argv[1], n_mol, n_ds, sys.argv[2]))
