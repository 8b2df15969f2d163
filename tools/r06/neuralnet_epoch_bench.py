"""What a user of deeprank_gnn_amd.NeuralNet gets end to end: `NeuralNet(database, GINet, ...).train(nepoch)` on a file of
synthetic SYN graphs (written here in the GraphStore .npz layout: node_data/*, edge_index, edge_data/dist, score/irmsd,
clustering/mcl/depth_{0,1} -- the reference's tree, Graph.py:61-139), us per mini-batch over whole train() calls (every
host-side piece of an epoch included: shuffle, the epoch's bookkeeping, the progress line), then cProfile of one train() call.
    python tools/r06/neuralnet_epoch_bench.py [graphs] [batch] [epochs]"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.NeuralNet import NeuralNet                # noqa: E402
from deeprank_gnn_amd.ginet import GINet                        # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
E = int(sys.argv[3]) if len(sys.argv) > 3 else 20
VAL = len(sys.argv) > 4 and sys.argv[4] == "validate"        # percent [0.8, 0.2] + train(validate=True): a validation pass per epoch
tmp = tempfile.mkdtemp()
db = os.path.join(tmp, "syn.npz")
synth.save_store(db, G)
torch.manual_seed(0)
quiet = io.StringIO()
with redirect_stdout(quiet):
    nn = NeuralNet(db, GINet, node_feature=["feat"], edge_feature=["dist"], target="irmsd", batch_size=B,
                   percent=[0.8, 0.2] if VAL else [1.0, 0.0], outdir=tmp)
    nn.train(nepoch=3, validate=VAL, save_model=None, hdf5=None)              # warm-up (upload, topology cache, allocations)
torch.cuda.synchronize()
nb = (len(nn.train_index) + B - 1) // B + ((len(nn.valid_index) + B - 1) // B if VAL else 0)
for rep in range(3):
    with redirect_stdout(quiet):
        t0 = time.perf_counter()
        nn.train(nepoch=E, validate=VAL, save_model=None, hdf5=None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("NeuralNet.train(%d epochs) over %d graphs, batch %d (%d mini-batches per epoch), cached=%s: %.2f us per mini-batch (%.2f M graphs/s)"
          % (E, G, B, nb, nn._use_cache(nn._resident(nn.dataset)), dt / (E * nb) * 1e6, E * G / dt / 1e6))
pr = cProfile.Profile()
with redirect_stdout(quiet):
    pr.enable()
    nn.train(nepoch=E, validate=VAL, save_model=None, hdf5=None)
    pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16)
print(s.getvalue())
