// libvoltrix_hip.so -- the CSR row-gather kernel (include/voltrix_capi.h; voltrix/spmm_csr_kernels.hpp): C = A * B straight from a
// device CSR, fp32 / fp16 / bf16 rows, fp32 result.
#include <hip/hip_runtime.h>

#include "voltrix/spmm_csr_kernels.hpp"
#include "voltrix_capi.h"

extern "C" {

void voltrix_launch_spmm_csr_rows(void* indptr, void* indices, int num_rows, int embedding_dim, void* input, int dtype, void* output,
                                  int xcd_ranges, void* stream, int* return_code) {
  *return_code = voltrix::launch_spmm_csr_rows(static_cast<const int*>(indptr), static_cast<const int*>(indices), num_rows,
                                               embedding_dim, input, dtype, static_cast<float*>(output),
                                               static_cast<hipStream_t>(stream), xcd_ranges);
}

}  // extern "C"
