"""Per-kernel launch wrappers (reference voltrix/jit_kernels/__init__.py:1-4) + the fused GPU preprocess."""
from .bmat_swizzle import hmat_packed_swizzle_kernel
from .hmat_gem import hmat_gen_kernel
from .spmm import spmm_kernel
from .preprocess import preprocess_kernel
from .csr_fused import csr_fused_preprocess_kernel
from .tuner import jit_tuner
