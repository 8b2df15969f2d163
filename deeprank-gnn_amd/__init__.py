"""MI355X-native message-passing hot path of Deeprank-GNN (GINet / sGAT / FoutNet
convolutions + community pooling), behind the reference's own model API.

Importing the package never touches the GPU or the HIP library; the first call that
needs a kernel loads ``csrc/libdrgnn.so`` and raises if it (or a GPU) is missing --
there is no CPU fallback in the product path.
"""
from .data import Batch, Data, DataLoader  # noqa: F401
from .ginet import GINet  # noqa: F401
from .sGAT import sGAT  # noqa: F401
from .foutnet import FoutNet  # noqa: F401

__version__ = "0.1.0"
