#!/usr/bin/env python
"""Native .drgs container (tree mirror) -> HDF5 in the reference's schema (what Graph.nx2h5 writes, reference
Graph.py:61-139; what HDF5DataSet.load_one_graph reads, reference DataSet.py:231-366): one group per molecule, datasets
under their original paths (nodes, node_data/*, edge_index, edges, edge_data/*, internal_*, score/*, clustering/*/*).
Run where h5py exists:  /opt/conda/bin/python3.9 tools/native_to_hdf5.py in.drgs out.hdf5"""
import importlib.util
import os
import sys

import h5py

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("drgs_container", os.path.join(HERE, "..", "deeprank-gnn_amd", "container.py"))
container = importlib.util.module_from_spec(spec)
spec.loader.exec_module(container)


def convert(src, dst):
    meta, sections = container.read_container(src, prefix="tree/")
    with h5py.File(dst, "w") as f:
        for mol in meta.get("mols", []):
            f.require_group(mol)
        for name, arr in sections.items():
            f.create_dataset(name[len("tree/"):], data=arr)
        for grp, attrs in meta.get("attrs", {}).items():       # e.g. the epoch groups of a training export
            for k, v in attrs.items():
                f[grp].attrs[k] = v
    return len(sections)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    print("%d datasets -> %s" % (convert(sys.argv[1], sys.argv[2]), sys.argv[2]))
