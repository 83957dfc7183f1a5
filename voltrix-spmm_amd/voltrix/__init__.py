"""voltrix -- MI355X-native (gfx950 / CDNA4) drop-in for the Voltrix-SpMM operator surface.

Same public names as the reference package (voltrix/__init__.py:1-3): ``BLK_H``, ``BLK_W``,
``csr_preprocess``, ``spmm``, the four ``*_kernel`` wrappers, ``jit`` and the ``VOLTRIX_*`` flag names.
"""
from .project import *  # noqa: F401,F403
from .jit_kernels import *  # noqa: F401,F403
from .jit_kernels import (csr_fused_preprocess_kernel, hmat_gen_kernel, hmat_packed_swizzle_kernel, jit_tuner,
                          preprocess_kernel, spmm_kernel)
from .spmm import *  # noqa: F401,F403
from .spmm import (BLK_H, BLK_W, csr_preprocess, csr_preprocess_device, csr_preprocess_hybrid, spmm, spmm_two_level,
                   two_level_of)
from .hybrid import TwoLevelHandle
from .sidecar import copy_side_car, load_handle, save_handle, slim_handle
from .reorder import ReorderedHandle, csr_preprocess_reordered, permute_features, spmm_reordered, unpermute_output
from .weighted import WeightedHandle, csr_preprocess_weighted, spmm_weighted
from .weighted import update_values as update_edge_values
from .graphed import GraphedSpMM
from . import autograd, hybrid, jit, sidecar, utils

__version__ = "0.2.0"
