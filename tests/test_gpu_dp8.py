"""BASELINE configs[4] (GINet data parallel, batch 512 = 8 x 64) at its REAL size on the one MI355X of the test box:
8 ranks time-share the GPU (gloo carries the collectives; RCCL needs one GPU per rank), every rank owns the 64 SYN graphs
``64 r .. 64 r + 63`` and runs the benchmarked fused step on them, the flat gradient is all-reduced ONCE
(weighted 1 / world), Adam runs replicated.  SURVEY.md 8(e)'s criterion: DP-8 gradients == single-process gradients on the
same 512 graphs within 1e-5 relative.  The single process is rank 0 itself, stepping the union of the shards as one batch
of 512 (which takes the one-workgroup-per-graph layout)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD, PER_RANK = 8, 64


def _net():
    from deeprank_gnn_amd.ginet import GINet
    torch.manual_seed(5)
    net = GINet(32, 1, 1)
    net.dropout = 0.0
    return net


def _worker(rank, world, init_file, out_dir):
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    batch = synth.make_batch(rank * PER_RANK, PER_RANK).to(dev)
    tr = FusedTrainer(_net().to(dev), lr=0.01, task="reg")
    topo = Topology.from_batch(batch, need_weights=False)
    assert tr._can_fuse(topo, 32)
    assert tr.api.net_step_plan(tr.kind, 32, topo.max_nodes, topo.max_edges, topo.max_c0, tr.R, tr.H, tr.O, PER_RANK)[0] == 2
    tr.compute_gradients(batch, topo=topo)
    local = tr.flat_g.detach().cpu().numpy().copy()
    tr.all_reduce_gradients()                               # equal shards: 1 / world, ONE all-reduce of the flat buffer
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "g%d.npy" % rank), tr.flat_g.cpu().numpy())
    np.save(os.path.join(out_dir, "l%d.npy" % rank), local)
    tr.apply_update()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "p%d.npy" % rank), tr.flat_p.cpu().numpy())
    assert tr.faults() == 0
    if rank == 0:
        # the single-process reference on the union of the shards (512 graphs, ids 0 .. 511)
        ref = FusedTrainer(_net().to(dev), lr=0.01, task="reg")
        union = synth.make_batch(0, world * PER_RANK).to(dev)
        ref.compute_gradients(union)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "ref_g.npy"), ref.flat_g.cpu().numpy())
        ref.apply_update()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "ref_p.npy"), ref.flat_p.cpu().numpy())
        assert ref.faults() == 0
    dist.barrier()
    dist.destroy_process_group()


def test_dp512_gradients_on_eight_ranks_equal_the_single_process():
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(WORLD, init_file, tmp), nprocs=WORLD, join=True)
        g = [np.load(os.path.join(tmp, "g%d.npy" % r)) for r in range(WORLD)]
        loc = [np.load(os.path.join(tmp, "l%d.npy" % r)) for r in range(WORLD)]
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(WORLD)]
        ref_g, ref_p = np.load(os.path.join(tmp, "ref_g.npy")), np.load(os.path.join(tmp, "ref_p.npy"))
    for r in range(1, WORLD):
        np.testing.assert_array_equal(g[0], g[r])       # every rank holds the same reduced gradient ...
        np.testing.assert_array_equal(p[0], p[r])       # ... and the same parameters after Adam (params_in_sync)
    assert all(not np.array_equal(loc[0], loc[r]) for r in range(1, WORLD))     # the shards really differ
    scale = float(np.abs(ref_g).max())
    np.testing.assert_allclose(g[0], ref_g, rtol=1e-5, atol=1e-5 * scale)       # 1e-5 relative
    np.testing.assert_allclose(p[0], ref_p, rtol=1e-5, atol=2e-6)


def test_bench_dp_selftest_two_ranks_over_gloo():
    """bench.py --dp-selftest: 3 eager + 3 recorded steps on every rank, gradients vs rank 0's recompute of the union shard
    and parameters in sync -- here with 2 ranks on the one GPU over gloo (the multi-GPU run uses RCCL the same way)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dp-selftest", "--steps", "20",
           "--warmup", "4", "--min-seconds", "0.2", "--no-cpu-baseline", "--epoch-graphs", "0"]
    run = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    line = [ln for ln in run.stdout.decode().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    st = res["config"]["dp_selftest"]
    assert st["ok"] and st["ranks"] == 2 and len(st["steps"]) >= 6
    assert all(s["params_in_sync"] and s["grad_max_rel_err"] <= 1e-5 for s in st["steps"]), st
    assert res["config"]["params_in_sync"] is True and res["n_gpus"] == 2
