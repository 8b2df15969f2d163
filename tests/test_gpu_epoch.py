"""Native epoch loop (drgnn_train_epoch) on the MI355X against stepping host-collated mini-batches one by one."""
import pytest

from collate_check import ragged_graphs
from epoch_check import check_epoch
from helpers import fixture_graphs
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Net,task,bs", [(GINet, "reg", 4), (sGAT, "reg", 3), (FoutNet, "reg", 9), (GINet, "class", 2)])
def test_epoch_ragged(Net, task, bs):
    check_epoch(Net, ragged_graphs(11, 12), 12, task, "cuda", bs)


def test_epoch_fixture():
    check_epoch(GINet, fixture_graphs(), 28, "reg", "cuda", 4)


@pytest.mark.parametrize("Net", [GINet, sGAT, FoutNet])
def test_epoch_full_size(Net):
    """SYN graphs at BASELINE size, 200 graphs in mini-batches of 64 (ragged last batch), two epochs."""
    import deeprank_gnn_amd.synthetic as synth
    check_epoch(Net, [synth.make_graph(i) for i in range(200)], 32, "reg", "cuda", 64)
