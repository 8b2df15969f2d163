"""graclus / normalized_cut / max_pool, kernels emulated on the CPU."""
import pytest

from emu_api import emu
from graclus_check import check_custom_net_recipe, check_graclus


@pytest.mark.parametrize("seed", [0, 1])
def test_graclus_matches_oracle(seed):
    check_graclus("cpu", api=emu(), seed=seed)


def test_readme_custom_net_recipe():
    check_custom_net_recipe("cpu", api=emu())
