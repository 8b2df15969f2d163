"""The two-workgroup GINet layout (branch workgroups that wait for each other's readout, drgnn_step3.h / drgnn_step.h) under
REAL concurrency on one GPU: two processes step it at the same time, no barrier between them, 2 000
steps each.  The layout is only launched when every workgroup of the launch is resident on an EXCLUSIVE device
(drgnn_net_step_plan); two processes sharing the GPU break that assumption, so this is the test of what then happens: the
kernels' bounded waits must never expire (faults() == 0) and every process must get the numbers of a solo run, bit for bit.
DRGNN_SHARED_GPU=1 (or DRGNN_RESIDENT_CUS) is the documented way out for deployments that share a device: it turns the
waiting layouts off; the second test checks that switch."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
STEPS = 2000


def _run(n_steps):
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = GINet(32, 1, 1).to(dev)
    tr = FusedTrainer(net, lr=1e-3, task="reg", seed=9)
    batch = synth.make_batch(0, 64).to(dev)
    topos = [Topology.from_batch(batch, need_weights=False), Topology.from_batch(batch, need_weights=False, build=False)]
    wgs = tr.api.net_step_plan(tr.kind, 32, topos[0].max_nodes, topos[0].max_edges, topos[0].max_c0, tr.R, tr.H, tr.O, 64, 64)[0]
    for it in range(n_steps):
        tr.train_step(batch, topo=topos[it & 1], next_topo=topos[1 - (it & 1)])
    torch.cuda.synchronize()
    return wgs, tr.faults(), float(tr.loss), tr.flat_p.cpu().numpy()


def _worker(rank, out_dir, n_steps):
    torch.set_num_threads(2)
    wgs, faults, loss, p = _run(n_steps)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), p)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as fh:
        fh.write("%d %d %r" % (wgs, faults, loss))


def test_two_processes_step_the_two_workgroup_layout_concurrently():
    wgs, faults, loss, solo = _run(STEPS)
    assert wgs == 2 and faults == 0 and np.isfinite(loss)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(tmp, STEPS), nprocs=2, join=True)
        for r in range(2):
            w, f, l = open(os.path.join(tmp, "r%d.txt" % r)).read().split()
            assert int(w) == 2 and int(f) == 0, (w, f, l)
            np.testing.assert_array_equal(np.load(os.path.join(tmp, "p%d.npy" % r)), solo)


def _worker_shared(rank, out_dir):
    os.environ["DRGNN_SHARED_GPU"] = "1"
    _worker(rank, out_dir, 50)


def test_shared_gpu_switch_turns_the_waiting_layouts_off():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_shared, args=(tmp,), nprocs=1, join=True)
        w, f, l = open(os.path.join(tmp, "r0.txt")).read().split()
        assert int(w) == 1 and int(f) == 0 and np.isfinite(float(l))
