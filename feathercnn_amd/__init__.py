"""feathercnn_amd -- MI355X (gfx950) native convolution hot path behind FeatherCNN's booster operator API.

The product is the C-ABI shared library ``libfeather_hip.so`` (declared in ``include/feather_hip/feather_hip.h``,
C++ host class in ``include/booster/booster.h``).  This package is the Python host-side mirror of that same
interface (``ConvParam`` / ``ConvBooster`` with the reference's method names) used by the tests and the bench;
PyTorch only provides device memory, streams and ``torch.distributed`` plumbing.

There is NO CPU fallback: importing works anywhere, but every compute call requires the HIP library and a GPU
and fails loudly otherwise.
"""
from .booster import (ALGO_NAMES, DEPTHWISE, IM2COL, NAIVE, SGECONV, WINOGRADF23, WINOGRADF63, WINOGRADF63FUSED,
                      ConvBooster, ConvLayer, ConvParam, FeatherHipError, None_, ReLU)
from ._lib import lib_path, load_library

__all__ = ["ConvParam", "ConvBooster", "ConvLayer", "FeatherHipError", "load_library", "lib_path", "NAIVE", "IM2COL",
           "SGECONV", "DEPTHWISE", "WINOGRADF63", "WINOGRADF63FUSED", "WINOGRADF23", "None_", "ReLU", "ALGO_NAMES"]
