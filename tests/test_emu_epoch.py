"""Native epoch loop (drgnn_train_epoch), kernels emulated on the CPU."""
import pytest

from collate_check import ragged_graphs
from emu_api import emu
from epoch_check import check_epoch
from helpers import fixture_graphs, NODE_FEATURES
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet


@pytest.mark.parametrize("Net,task,bs", [(GINet, "reg", 4), (sGAT, "reg", 3), (FoutNet, "reg", 9), (GINet, "class", 2),
                                         (FoutNet, "class", 5)])
def test_epoch_ragged(Net, task, bs):
    check_epoch(Net, ragged_graphs(11, 12), 12, task, "cpu", bs, api=emu())


def test_epoch_fixture():
    check_epoch(GINet, fixture_graphs(), 28, "reg", "cpu", 4, api=emu())


@pytest.mark.parametrize("Net,task,bs", [(GINet, "reg", 4), (sGAT, "reg", 3), (FoutNet, "class", 5)])
def test_epoch_cached_topology(Net, task, bs):
    """Declared cached-topology mode (topology of every graph built once at upload, mini-batch = list of graph numbers)."""
    check_epoch(Net, ragged_graphs(11, 12), 12, task, "cpu", bs, api=emu(), cached=True)


@pytest.mark.parametrize("cached", [False, True])
def test_long_epochs_are_enqueued_in_pieces_with_identical_results(cached, monkeypatch):
    """An epoch longer than FusedTrainer.EPOCH_CHUNK mini-batches goes to the native loop in pieces (the host stays at most
    two pieces ahead of the device): same losses / predictions / parameters as one call, bit for bit."""
    from deeprank_gnn_amd.trainer import FusedTrainer
    monkeypatch.setattr(FusedTrainer, "EPOCH_CHUNK", 2)       # 11 graphs, batch 2 -> 6 mini-batches -> 3 pieces
    check_epoch(GINet, ragged_graphs(11, 12), 12, "reg", "cpu", 2, api=emu(), cached=cached)
