// Vendor baseline worth beating (round 6): rocSPARSE's generic SpMM with EVERY CSR algorithm the library
// offers, and a plain CSR row-gather kernel ("no format": what a straightforward HIP kernel does on the same CSR).
// Plain HIP + rocSPARSE, no torch: a C-ABI library (harness/bm_rocsparse.py drives it through ctypes with device pointers) and,
// built with -DBM_ROCSPARSE_MAIN, a standalone program on a seeded random CSR.
//
// Role in the reference: bench/bm_sparse.py:6-52 times cuSPARSE's CSR SpMM (through torch.sparse) -- buffer sizing and analysis
// outside the timed loop (bm_sparse.py:20-45); every number the reference publishes is a speed-up over that (bench/plot.py:105-123).
// Here: rocsparse_spmm stages buffer_size + preprocess run ONCE outside the timed loop, only stage compute is timed.
//
// Bench infrastructure: nothing in the product links or loads this file.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return -1000 - (int)e_; } } while (0)
#define RS_OK(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { return (int)s_; } } while (0)

namespace {

// ---- the "no format" floor: one group of F / VEC lanes per row, 16-byte loads of B, fp32 accumulate, UNROLL edges in flight ----
template <typename T> struct Vec16;
template <> struct Vec16<__half> { static constexpr int N = 8; };
template <> struct Vec16<float> { static constexpr int N = 4; };

template <typename T> __device__ inline void accumulate(float* acc, const uint4& raw);
template <> __device__ inline void accumulate<__half>(float* acc, const uint4& raw) {
    const __half2* h = reinterpret_cast<const __half2*>(&raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float2 f = __half22float2(h[i]);
        acc[2 * i] += f.x;
        acc[2 * i + 1] += f.y;
    }
}
template <> __device__ inline void accumulate<float>(float* acc, const uint4& raw) {
    const float* f = reinterpret_cast<const float*>(&raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += f[i];
}

template <typename T, int UNROLL>
__global__ void __launch_bounds__(256) csr_row_gather_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                             const T* __restrict__ b, float* __restrict__ c, int num_rows,
                                                             int num_feats, int lanes_per_row) {
    constexpr int V = Vec16<T>::N;
    const int64_t thread = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = thread / lanes_per_row;
    const int lane = (int)(thread % lanes_per_row);
    if (row >= num_rows) return;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.0f;
    int e = indptr[row];
    const int end = indptr[row + 1];
    const int64_t col0 = (int64_t)lane * V;
    for (; e + UNROLL <= end; e += UNROLL) {
        uint4 raw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            raw[u] = *reinterpret_cast<const uint4*>(b + (int64_t)indices[e + u] * num_feats + col0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) accumulate<T>(acc, raw[u]);
    }
    for (; e < end; ++e) {
        uint4 raw = *reinterpret_cast<const uint4*>(b + (int64_t)indices[e] * num_feats + col0);
        accumulate<T>(acc, raw);
    }
    float4* out = reinterpret_cast<float4*>(c + row * num_feats + col0);
#pragma unroll
    for (int i = 0; i < V / 4; ++i) out[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { hipEventCreate(&a); hipEventCreate(&b); }
    ~Timer() { hipEventDestroy(a); hipEventDestroy(b); }
};

// one timed batch of `iters` back-to-back calls, or -- flush_bytes > 0, the reference's protocol (voltrix/utils.py:277-281,
// bm_voltrix.py:36) -- every call after a write of flush_bytes, its own event pair, mean returned
template <typename F> int timed(F&& call, int warmup, int iters, void* flush, size_t flush_bytes, hipStream_t stream, float* ms) {
    Timer t;
    for (int i = 0; i < warmup; ++i) { int rc = call(); if (rc) return rc; }
    HIP_OK(hipStreamSynchronize(stream));
    if (flush_bytes == 0) {
        HIP_OK(hipEventRecord(t.a, stream));
        for (int i = 0; i < iters; ++i) { int rc = call(); if (rc) return rc; }
        HIP_OK(hipEventRecord(t.b, stream));
        HIP_OK(hipEventSynchronize(t.b));
        HIP_OK(hipEventElapsedTime(ms, t.a, t.b));
        *ms /= iters;
        return 0;
    }
    float total = 0.0f;
    for (int i = 0; i < iters; ++i) {
        HIP_OK(hipMemsetAsync(flush, i & 1, flush_bytes, stream));
        HIP_OK(hipEventRecord(t.a, stream));
        int rc = call(); if (rc) return rc;
        HIP_OK(hipEventRecord(t.b, stream));
        HIP_OK(hipEventSynchronize(t.b));
        float one = 0.0f;
        HIP_OK(hipEventElapsedTime(&one, t.a, t.b));
        total += one;
    }
    *ms = total / iters;
    return 0;
}

}  // namespace

extern "C" {

// dtype: 0 = fp32 values / B, 1 = fp16 values / B (C and compute fp32 either way).  alg: rocsparse_spmm_alg as an int
// (1 csr, 4 csr_row_split, 5 csr_nnz_split (= csr_merge), 9 csr_merge_path, 0 default).  `values` = the matrix values in the
// operand's dtype (ones for the binary adjacency).  Returns 0 and *ms = mean milliseconds of stage compute, or the
// rocsparse_status / -1000 - hipError of the failing call (3 = not implemented for this combination).
int bm_rocsparse_spmm(const void* indptr, const void* indices, const void* values, int64_t num_rows, int64_t num_cols,
                      int64_t nnz, int num_feats, const void* b, void* c, int dtype, int alg, int warmup, int iters,
                      void* flush, size_t flush_bytes, void* stream_v, float* ms, size_t* buffer_bytes, float* preprocess_ms) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    rocsparse_handle handle;
    RS_OK(rocsparse_create_handle(&handle));
    RS_OK(rocsparse_set_stream(handle, stream));
    rocsparse_datatype dt = dtype == 1 ? rocsparse_datatype_f16_r : rocsparse_datatype_f32_r;
    rocsparse_spmat_descr a_descr;
    rocsparse_dnmat_descr b_descr, c_descr;
    RS_OK(rocsparse_create_csr_descr(&a_descr, num_rows, num_cols, nnz, const_cast<void*>(indptr), const_cast<void*>(indices),
                                     const_cast<void*>(values), rocsparse_indextype_i32, rocsparse_indextype_i32,
                                     rocsparse_index_base_zero, dt));
    RS_OK(rocsparse_create_dnmat_descr(&b_descr, num_cols, num_feats, num_feats, const_cast<void*>(b), dt, rocsparse_order_row));
    RS_OK(rocsparse_create_dnmat_descr(&c_descr, num_rows, num_feats, num_feats, c, rocsparse_datatype_f32_r, rocsparse_order_row));
    const float alpha = 1.0f, beta = 0.0f;
    const rocsparse_spmm_alg algorithm = static_cast<rocsparse_spmm_alg>(alg);
    size_t bytes = 0;
    int rc = (int)rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, a_descr, b_descr, &beta, c_descr,
                                 rocsparse_datatype_f32_r, algorithm, rocsparse_spmm_stage_buffer_size, &bytes, nullptr);
    void* buffer = nullptr;
    if (rc == 0) {
        if (hipMalloc(&buffer, std::max<size_t>(bytes, 16)) != hipSuccess) rc = -1002;
    }
    if (rc == 0) {
        Timer t;
        hipEventRecord(t.a, stream);
        rc = (int)rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, a_descr, b_descr, &beta, c_descr,
                                 rocsparse_datatype_f32_r, algorithm, rocsparse_spmm_stage_preprocess, &bytes, buffer);
        hipEventRecord(t.b, stream);
        hipEventSynchronize(t.b);
        if (preprocess_ms) hipEventElapsedTime(preprocess_ms, t.a, t.b);
    }
    if (rc == 0) {
        auto call = [&]() -> int {
            return (int)rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, a_descr, b_descr, &beta,
                                       c_descr, rocsparse_datatype_f32_r, algorithm, rocsparse_spmm_stage_compute, &bytes, buffer);
        };
        rc = timed(call, warmup, iters, flush, flush_bytes, stream, ms);
    }
    if (buffer_bytes) *buffer_bytes = bytes;
    if (buffer) hipFree(buffer);
    rocsparse_destroy_dnmat_descr(b_descr);
    rocsparse_destroy_dnmat_descr(c_descr);
    rocsparse_destroy_spmat_descr(a_descr);
    rocsparse_destroy_handle(handle);
    return rc;
}

// The plain CSR row-gather kernel (binary adjacency: values implicit).  num_feats must be a multiple of 8 (fp16) / 4 (fp32)
// and F / that a divisor of 256.  unroll in {1, 4, 8}.
int bm_csr_row_gather(const void* indptr, const void* indices, int64_t num_rows, int num_feats, const void* b, void* c, int dtype,
                      int unroll, int warmup, int iters, void* flush, size_t flush_bytes, void* stream_v, float* ms) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int vec = dtype == 1 ? 8 : 4;
    if (num_feats % vec) return -1;
    const int lanes = num_feats / vec;
    if (lanes > 256 || 256 % lanes) return -2;
    const int64_t threads = num_rows * lanes;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    auto call = [&]() -> int {
        const int* ip = static_cast<const int*>(indptr);
        const int* ix = static_cast<const int*>(indices);
        float* out = static_cast<float*>(c);
#define LAUNCH(T, U) hipLaunchKernelGGL((csr_row_gather_kernel<T, U>), dim3(blocks), dim3(256), 0, stream, ip, ix, static_cast<const T*>(b), out, (int)num_rows, num_feats, lanes)
        if (dtype == 1) { if (unroll >= 8) LAUNCH(__half, 8); else if (unroll >= 4) LAUNCH(__half, 4); else LAUNCH(__half, 1); }
        else            { if (unroll >= 8) LAUNCH(float, 8); else if (unroll >= 4) LAUNCH(float, 4); else LAUNCH(float, 1); }
#undef LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : -3;
    };
    return timed(call, warmup, iters, flush, flush_bytes, stream, ms);
}

}  // extern "C"

#ifdef BM_ROCSPARSE_MAIN
// standalone: bm_rocsparse <rows> <mean degree> <F> [fp16=1]  -- uniform random CSR, every algorithm + the row-gather kernel
int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 65536;
    const int deg = argc > 2 ? std::atoi(argv[2]) : 32;
    const int feats = argc > 3 ? std::atoi(argv[3]) : 128;
    const int dtype = argc > 4 ? std::atoi(argv[4]) : 1;
    std::vector<int> indptr(n + 1), indices;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int64_t r = 0; r < n; ++r) {
        std::vector<int> row(deg);
        for (auto& v : row) v = (int)(rnd() % n);
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        indptr[r] = (int)indices.size();
        indices.insert(indices.end(), row.begin(), row.end());
    }
    indptr[n] = (int)indices.size();
    const int64_t nnz = indices.size();
    const size_t es = dtype == 1 ? 2 : 4;
    void *d_ip, *d_ix, *d_val, *d_b, *d_c;
    hipMalloc(&d_ip, (n + 1) * 4); hipMalloc(&d_ix, nnz * 4); hipMalloc(&d_val, nnz * es);
    hipMalloc(&d_b, n * feats * es); hipMalloc(&d_c, n * feats * 4);
    hipMemcpy(d_ip, indptr.data(), (n + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ix, indices.data(), nnz * 4, hipMemcpyHostToDevice);
    if (dtype == 1) { std::vector<__half> ones(nnz, __float2half(1.0f)); hipMemcpy(d_val, ones.data(), nnz * 2, hipMemcpyHostToDevice); }
    else { std::vector<float> ones(nnz, 1.0f); hipMemcpy(d_val, ones.data(), nnz * 4, hipMemcpyHostToDevice); }
    hipMemset(d_b, 0, n * feats * es);
    const int algs[] = {1, 4, 5, 9};
    const char* names[] = {"csr", "csr_row_split", "csr_nnz_split", "csr_merge_path"};
    for (int i = 0; i < 4; ++i) {
        float ms = 0, pre = 0; size_t bytes = 0;
        int rc = bm_rocsparse_spmm(d_ip, d_ix, d_val, n, n, nnz, feats, d_b, d_c, dtype, algs[i], 3, 10, nullptr, 0, nullptr, &ms, &bytes, &pre);
        std::printf("rocsparse_spmm_alg_%-15s rc=%d %.4f ms (buffer %zu B, preprocess %.3f ms)\n", names[i], rc, ms, bytes, pre);
    }
    for (int u : {1, 4, 8}) {
        float ms = 0;
        int rc = bm_csr_row_gather(d_ip, d_ix, n, feats, d_b, d_c, dtype, u, 3, 10, nullptr, 0, nullptr, &ms);
        std::printf("csr_row_gather unroll %d rc=%d %.4f ms\n", u, rc, ms);
    }
    return 0;
}
#endif
