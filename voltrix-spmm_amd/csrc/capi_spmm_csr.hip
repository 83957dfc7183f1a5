// libvoltrix_hip.so -- the CSR row-gather kernel (include/voltrix_capi.h; voltrix/spmm_csr_kernels.hpp): C = A * B straight from a
// device CSR, fp32 / fp16 / bf16 rows, fp32 result, binary or with fp32 edge values; the value-plane scatter of update_values.
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

void voltrix_launch_spmm_csr_rows_weighted(void* indptr, void* indices, void* values, int num_rows, int embedding_dim, void* input, int dtype,
                                           void* output, int xcd_ranges, void* stream, int* return_code) {
  *return_code = values == nullptr ? voltrix::kErrBadShape
                                   : voltrix::launch_spmm_csr_rows(static_cast<const int*>(indptr), static_cast<const int*>(indices), num_rows,
                                                                   embedding_dim, input, dtype, static_cast<float*>(output),
                                                                   static_cast<hipStream_t>(stream), xcd_ranges,
                                                                   static_cast<const float*>(values));
}

void voltrix_launch_scatter_values(void* values, void* slots, void* plane, int64_t count, int dtype, void* stream, int* return_code) {
  *return_code = voltrix::scatter_values(static_cast<const float*>(values), static_cast<const long long*>(slots), plane, count, dtype,
                                         static_cast<hipStream_t>(stream));
}

}  // extern "C"
