"""Data-parallel path, world_size 2 over gloo on CPU (kernels: host-emulation build).

Each rank owns a contiguous shard of the graphs, runs compute_gradients on it, the flat
gradient buffer is all-reduced ONCE, Adam runs replicated.  Checked against a single
process training on the union of the shards (DP gradients == single-process
gradients on the same graphs)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _graphs(n):
    import deeprank_gnn_amd.synthetic as synth
    return [synth.make_graph(i, n_nodes=14, n_pairs=20, n_feat=8, n_c1=2, n_internal=6) for i in range(n)]


def _make_net(name):
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.sGAT import sGAT
    torch.manual_seed(3)
    net = {"GINet": GINet, "sGAT": sGAT}[name](8, 1, 1)
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    return net


def _worker(rank, world, init_file, net_name, sizes, out_dir):
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.parallel import shard_range
    from deeprank_gnn_amd.trainer import FusedTrainer
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    graphs = _graphs(sum(sizes))
    lo = sum(sizes[:rank])
    batch = Batch.from_data_list(graphs[lo:lo + sizes[rank]])
    tr = FusedTrainer(_make_net(net_name), lr=0.01, api=emu())
    for _ in range(2):
        tr.train_step(batch, n_global=sum(sizes))       # reduce -> ONE all-reduce -> Adam
    np.save(os.path.join(out_dir, "p%d.npy" % rank), tr.flat_p.numpy())
    np.save(os.path.join(out_dir, "g%d.npy" % rank), tr.flat_g.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("net_name,sizes", [("GINet", [6, 6]), ("sGAT", [7, 4])])
def test_two_rank_training_matches_single_process(net_name, sizes):
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.trainer import FusedTrainer
    emu()                                               # build the emulation library once, here
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(2, init_file, net_name, sizes, tmp), nprocs=2, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(2)]
        g = [np.load(os.path.join(tmp, "g%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(p[0], p[1])           # replicas stay bit-identical
    np.testing.assert_array_equal(g[0], g[1])
    tr = FusedTrainer(_make_net(net_name), lr=0.01, api=emu())
    union = Batch.from_data_list(_graphs(sum(sizes)))
    for _ in range(2):
        tr.train_step(union)
    scale = max(1.0, float(np.abs(tr.flat_g.numpy()).max()))
    np.testing.assert_allclose(g[0], tr.flat_g.numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(p[0], tr.flat_p.numpy(), rtol=1e-4, atol=1e-5)


def _worker8(rank, world, init_file, out_dir):
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.trainer import FusedTrainer
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    graphs = _graphs(8 * world)
    batch = Batch.from_data_list(graphs[8 * rank:8 * rank + 8])
    tr = FusedTrainer(_make_net("GINet"), lr=0.01, api=emu())
    tr.compute_gradients(batch)
    np.save(os.path.join(out_dir, "l%d.npy" % rank), tr.flat_g.numpy().copy())     # this rank's shard gradient
    tr.all_reduce_gradients()                                                       # equal shards: 1 / world
    np.save(os.path.join(out_dir, "g%d.npy" % rank), tr.flat_g.numpy().copy())
    tr.apply_update()
    np.save(os.path.join(out_dir, "p%d.npy" % rank), tr.flat_p.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gradients_match_single_process():
    """DP-8 gradients == single-process gradients on the same graphs (1e-5 relative).  8 ranks x 8
    graphs over gloo; the single process runs the union of the 64 graphs as one batch."""
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.trainer import FusedTrainer
    emu()
    world = 8
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker8, args=(world, init_file, tmp), nprocs=world, join=True)
        g = [np.load(os.path.join(tmp, "g%d.npy" % r)) for r in range(world)]
        loc = [np.load(os.path.join(tmp, "l%d.npy" % r)) for r in range(world)]
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(g[0], g[r])       # every rank holds the same reduced gradient ...
        np.testing.assert_array_equal(p[0], p[r])       # ... and the same parameters after Adam
    assert any(not np.array_equal(loc[0], loc[r]) for r in range(1, world))     # shards really differ
    tr = FusedTrainer(_make_net("GINet"), lr=0.01, api=emu())
    tr.compute_gradients(Batch.from_data_list(_graphs(8 * world)))
    ref = tr.flat_g.numpy()
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(g[0], ref, rtol=1e-5, atol=1e-5 * scale)
    tr.apply_update()
    np.testing.assert_allclose(p[0], tr.flat_p.numpy(), rtol=1e-5, atol=1e-6)


def _worker_epoch(rank, world, init_file, out_dir, cached):
    """Native epoch loop under data parallelism: per mini-batch gradient launches -> exchange callback (ONE all-reduce
    of the flat gradient) -> Adam, ragged shards weighted n_local / n_global."""
    from emu_api import emu
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    graphs = _graphs(14)
    rs = ResidentGraphSet(graphs, "cpu", api=emu())
    tr = FusedTrainer(_make_net("sGAT"), lr=0.01, api=emu(), seed=5)
    # global mini-batches of 6, 6, 2 graphs; rank 0 takes the first ceil(half) of each, rank 1 the rest
    order = list(range(14))
    mine, sizes = [], []
    for lo in range(0, 14, 6):
        chunk = order[lo:lo + 6]
        half = (len(chunk) + 1) // 2
        part = chunk[:half] if rank == 0 else chunk[half:]
        mine.append(part)
        sizes.append(len(chunk))
    local_bs = 3
    assert all(0 < len(p) <= local_bs for p in mine)
    for _ in range(2):
        # ragged local batches: one train_epoch call per mini-batch keeps the batch boundaries aligned across ranks
        for k, part in enumerate(mine):
            done = tr.train_epoch(rs, part, local_bs, cached=cached, dp_global_sizes=[sizes[k]])
            assert done is not None
    np.save(os.path.join(out_dir, "p%d.npy" % rank), tr.flat_p.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cached", [False, True])
def test_native_epoch_loop_under_data_parallel(cached):
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.trainer import FusedTrainer
    emu()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_epoch, args=(2, init_file, tmp, cached), nprocs=2, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(p[0], p[1])
    graphs = _graphs(14)
    tr = FusedTrainer(_make_net("sGAT"), lr=0.01, api=emu(), seed=5)
    for _ in range(2):
        for lo in range(0, 14, 6):
            tr.train_step(Batch.from_data_list(graphs[lo:lo + 6]))
    np.testing.assert_allclose(p[0], tr.flat_p.numpy(), rtol=1e-4, atol=1e-5)


def _worker_epoch_chunked(rank, world, init_file, out_dir):
    """ONE train_epoch call over five mini-batches with FusedTrainer.EPOCH_CHUNK = 2 (the native loop is entered three times)
    and a ragged LAST global mini-batch (4, 4, 4, 4, 3 graphs): the exchange's n_local / n_global weight must take the
    global size of mini-batch (piece offset + k) -- ADVICE r03."""
    from emu_api import emu
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    rs = ResidentGraphSet(_graphs(19), "cpu", api=emu())
    tr = FusedTrainer(_make_net("sGAT"), lr=0.01, api=emu(), seed=5)
    tr.EPOCH_CHUNK = 2
    mine = [g for lo in range(0, 19, 4) for g in range(lo, min(lo + 4, 19))[2 * rank:2 * rank + 2]]
    assert len(mine) == (10 if rank == 0 else 9)
    tr.train_epoch(rs, mine, 2, dp_global_sizes=[4, 4, 4, 4, 3])
    np.save(os.path.join(out_dir, "p%d.npy" % rank), tr.flat_p.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_native_epoch_under_data_parallel_weights_the_ragged_last_batch():
    from emu_api import emu
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.trainer import FusedTrainer
    emu()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_epoch_chunked, args=(2, init_file, tmp), nprocs=2, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(p[0], p[1])
    graphs = _graphs(19)
    tr = FusedTrainer(_make_net("sGAT"), lr=0.01, api=emu(), seed=5)
    for lo in range(0, 19, 4):
        tr.train_step(Batch.from_data_list(graphs[lo:lo + 4]))
    np.testing.assert_allclose(p[0], tr.flat_p.numpy(), rtol=1e-4, atol=1e-5)


def _worker_nn(rank, world, init_file, out_dir):
    from emu_api import emu
    from helpers import GOLDEN, NODE_FEATURES
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(os.path.join(GOLDEN, "1ATN_residue.drgs"), GINet, node_feature=NODE_FEATURES, edge_feature=['dist'],
                   target='irmsd', batch_size=4, percent=[1.0, 0.0], shuffle=True, outdir=out_dir, _api=emu(), device='cpu')
    nn.model.dropout = 0.0
    nn.train(nepoch=2, validate=False, save_model=None, hdf5=None)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), nn.trainer.flat_p.numpy())
    np.save(os.path.join(out_dir, "l%d.npy" % rank), np.asarray(nn.train_loss))
    dist.barrier()
    dist.destroy_process_group()


def test_neuralnet_trains_data_parallel_like_a_single_process(tmp_path):
    """deeprank_gnn_amd.NeuralNet under torch.distributed (2 ranks, gloo): batch_size is the GLOBAL mini-batch, every
    rank runs its shards through the native epoch loop; parameters and epoch losses equal the single-process run on the
    same (rank 0's) shuffled order."""
    from emu_api import emu
    from helpers import GOLDEN, NODE_FEATURES
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    emu()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_nn, args=(2, init_file, tmp), nprocs=2, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(2)]
        losses = [np.load(os.path.join(tmp, "l%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(p[0], p[1])
    np.testing.assert_array_equal(losses[0], losses[1])
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(os.path.join(GOLDEN, "1ATN_residue.drgs"), GINet, node_feature=NODE_FEATURES, edge_feature=['dist'],
                   target='irmsd', batch_size=4, percent=[1.0, 0.0], shuffle=True, outdir=str(tmp_path), _api=emu(),
                   device='cpu')
    nn.model.dropout = 0.0
    nn.train(nepoch=2, validate=False, save_model=None, hdf5=None)
    np.testing.assert_allclose(p[0], nn.trainer.flat_p.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(losses[0], nn.train_loss, rtol=1e-4)


def _worker_nn_modes(rank, world, init_file, out_dir, mode):
    """mode 'seeds': every rank seeds its RNGs differently (the replicas and the split must still be rank 0's);
    'small': a last global mini-batch smaller than the world; 'refuse': rank 1's native-loop probe says no."""
    from emu_api import emu
    from helpers import GOLDEN, NODE_FEATURES
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    seed = rank * 17 if mode == "seeds" else 0
    torch.manual_seed(seed)
    np.random.seed(seed)
    percent = [0.8, 0.2] if mode == "seeds" else [1.0, 0.0]
    nn = NeuralNet(os.path.join(GOLDEN, "1ATN_residue.drgs"), GINet, node_feature=NODE_FEATURES, edge_feature=['dist'],
                   target='irmsd', batch_size=4, percent=percent, shuffle=True, outdir=out_dir, _api=emu(), device='cpu')
    nn.model.dropout = 0.0
    calls = {"probe": 0, "run": 0}
    if mode == "refuse":
        orig = nn.trainer.train_epoch

        def train_epoch(*a, **kw):
            if kw.get("probe"):
                calls["probe"] += 1
                return None if rank == 1 else orig(*a, **kw)
            calls["run"] += 1
            return orig(*a, **kw)
        nn.trainer.train_epoch = train_epoch
    torch.manual_seed(0)               # (the epochs' shuffles: rank 0's order is broadcast anyway)
    nn.train(nepoch=2, validate=(mode == "seeds"), save_model='last' if mode == "seeds" else None,
             hdf5='train_data.drgs' if mode == "seeds" else None)
    if mode == "refuse":
        assert calls["probe"] == 2 and calls["run"] == 0, calls      # every rank left the native loop alone, together
    np.save(os.path.join(out_dir, "p%d.npy" % rank), nn.trainer.flat_p.numpy())
    np.save(os.path.join(out_dir, "l%d.npy" % rank), np.asarray(nn.train_loss))
    np.save(os.path.join(out_dir, "v%d.npy" % rank), np.asarray(nn.valid_index, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


def _single_process_reference(tmp_path):
    from emu_api import emu
    from helpers import GOLDEN, NODE_FEATURES
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(os.path.join(GOLDEN, "1ATN_residue.drgs"), GINet, node_feature=NODE_FEATURES, edge_feature=['dist'],
                   target='irmsd', batch_size=4, percent=[1.0, 0.0], shuffle=True, outdir=str(tmp_path), _api=emu(),
                   device='cpu')
    nn.model.dropout = 0.0
    torch.manual_seed(0)
    nn.train(nepoch=2, validate=False, save_model=None, hdf5=None)
    return nn


@pytest.mark.parametrize("mode,world", [("small", 3), ("refuse", 2)])
def test_neuralnet_data_parallel_fallback_is_collective_and_exact(tmp_path, mode, world):
    """(small) 10 graphs, global mini-batches of 4 -> the last one has 2 graphs for 3 ranks: the ranks step
    the SAME global mini-batches one by one, the rank without a graph contributes weight 0; (refuse) one rank's native
    loop refuses its shard: ALL ranks take the per-mini-batch path (agreed with an all-reduce), none is left inside the
    loop's collectives.  Both equal the single-process run on rank 0's order."""
    from emu_api import emu
    emu()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_nn_modes, args=(world, init_file, tmp, mode), nprocs=world, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(world)]
        losses = [np.load(os.path.join(tmp, "l%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(p[0], p[r])
        np.testing.assert_array_equal(losses[0], losses[r])
    nn = _single_process_reference(tmp_path)
    np.testing.assert_allclose(p[0], nn.trainer.flat_p.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(losses[0], nn.train_loss, rtol=1e-4)


def test_neuralnet_data_parallel_replicas_start_as_one_model():
    """ranks seeded differently still train ONE model (rank 0's parameters / optimiser state / train-validation
    split are broadcast at construction); checkpoints are written by rank 0 only, exports of ranks > 0 carry a suffix."""
    from emu_api import emu
    emu()
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_nn_modes, args=(2, init_file, tmp, "seeds"), nprocs=2, join=True)
        p = [np.load(os.path.join(tmp, "p%d.npy" % r)) for r in range(2)]
        v = [np.load(os.path.join(tmp, "v%d.npy" % r)) for r in range(2)]
        files = sorted(os.listdir(tmp))
    np.testing.assert_array_equal(p[0], p[1])
    np.testing.assert_array_equal(v[0], v[1])
    assert len(v[0]) == 2
    assert len([f for f in files if f.endswith(".pth.tar")]) == 1, files
    assert "train_data.drgs" in files and "train_data.rank1.drgs" in files, files


def test_shard_range_covers_everything():
    from deeprank_gnn_amd.parallel import shard_range
    for n in (0, 1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
