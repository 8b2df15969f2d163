"""Records the LAYOUT (group / dataset names, dtypes kinds, group attributes) of the reference-held export file
tests/data/train_ref/train_data.hdf5 as a small golden json.  Run in the build container:
    /opt/conda/bin/python3.9 tests/golden/gen/export_train_ref_layout.py"""
import json
import os

import h5py

SRC = "/root/reference/tests/data/train_ref/train_data.hdf5"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "train_ref_layout.json")
out = {"source": "tests/data/train_ref/train_data.hdf5", "groups": {}, "datasets": {}}
with h5py.File(SRC, "r") as f:
    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            out["datasets"][name] = {"kind": "S" if obj.dtype.kind in "OSU" else obj.dtype.kind, "ndim": obj.ndim}
        else:
            out["groups"][name] = sorted(str(k) for k in obj.attrs.keys())
    f.visititems(visit)
json.dump(out, open(DST, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
