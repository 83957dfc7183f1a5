from .spmm import BLK_H, BLK_W
from .spmm import (
    csr_preprocess,
    csr_preprocess_hybrid,
    spmm,
)
