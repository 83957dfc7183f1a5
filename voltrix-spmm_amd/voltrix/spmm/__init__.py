from .spmm import BLK_H, BLK_W
from .spmm import (
    csr_preprocess,
    csr_preprocess_device,
    csr_preprocess_hybrid,
    spmm,
    spmm_two_level,
    two_level_of,
)
