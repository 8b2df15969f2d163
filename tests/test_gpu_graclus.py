"""graclus / normalized_cut / max_pool on the MI355X."""
import pytest

from graclus_check import check_custom_net_recipe, check_graclus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1])
def test_graclus_matches_oracle(seed):
    check_graclus("cuda", seed=seed)


def test_readme_custom_net_recipe():
    check_custom_net_recipe("cuda")
