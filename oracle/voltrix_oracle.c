/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library (oracle/_build/libvoltrix_oracle.so).  The product never links it.
 *
 * Each function restates one reference function (paths relative to the reference
 * checkout, voltrix/include/voltrix/):
 *   oracle_preprocess           bmat_kernels.cuh:264-320 (+ inplace_deduplication :248-262)
 *   oracle_hmat_gen             bmat_kernels.cuh:21-111
 *   oracle_hmat_packed_swizzle  bmat_kernels.cuh:151-193
 *   oracle_spmm_blocked         spmm_kernels.cuh:1632-1716 (a_frag bit test :1632-1644,
 *                               cvt.rna.tf32 :1642,1671, mma + fp32 accumulate :1647-1681,
 *                               row-major store :1687-1712)
 *   oracle_spmm_csr             tests/test_spmm.py:24-29,79-80 (csr(ones) @ feat), fp64 accumulate
 *
 * Pinning: see oracle/oracle_np.py header and DESIGN.md.  The reference's native
 * sources need nvcc + the CUDA runtime headers + PTX and cannot be built here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLK_H 16 /* traits.h:6 */
#define BLK_W 8  /* traits.h:7 */

static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return (x > y) - (x < y);
}

/* bmat_kernels.cuh:264-320.  Returns total TC blocks (block_counter, :309). */
int64_t oracle_preprocess(const int32_t *edge_list, const int32_t *node_pointer, int32_t num_nodes,
                          int32_t *block_partition, int32_t *edge_to_column, int32_t *edge_to_row,
                          int32_t *pointer1) {
  int64_t block_counter = 0;
  for (int32_t nid = 0; nid < num_nodes; nid++) /* :273-276 */
    for (int32_t eid = node_pointer[nid]; eid < node_pointer[nid + 1]; eid++) edge_to_row[eid] = nid;

  int32_t num_windows = (num_nodes + BLK_H - 1) / BLK_H;
  for (int32_t w = 0; w < num_windows; w++) { /* :279-308 */
    int32_t iter = w * BLK_H;
    int32_t lo = node_pointer[iter];
    int32_t end_row = iter + BLK_H < num_nodes ? iter + BLK_H : num_nodes;
    int32_t hi = node_pointer[end_row];
    int32_t n = hi - lo;
    if (n == 0) {
      /* :252 inserts array[0] even for length 0 -> map.size()==1 -> one empty TC block */
      block_partition[w] = 1;
      block_counter += 1;
      continue;
    }
    uint32_t *nb = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    memcpy(nb, edge_list + lo, (size_t)n * sizeof(uint32_t));
    qsort(nb, (size_t)n, sizeof(uint32_t), cmp_u32); /* :288 thrust::sort */
    int32_t u = 1;                                   /* :248-262 */
    for (int32_t i = 1; i < n; i++)
      if (nb[i] != nb[i - 1]) nb[u++] = nb[i];
    block_partition[w] = (u + BLK_W - 1) / BLK_W; /* :298-299 */
    block_counter += block_partition[w];
    for (int32_t e = lo; e < hi; e++) { /* :304-307 */
      uint32_t key = (uint32_t)edge_list[e];
      int32_t a = 0, b = u - 1;
      while (a < b) {
        int32_t m = (a + b) >> 1;
        if (nb[m] < key) a = m + 1; else b = m;
      }
      edge_to_column[e] = a;
    }
    free(nb);
  }
  pointer1[0] = 0; /* :312-319 */
  for (int32_t w = 0; w < num_windows; w++) pointer1[w + 1] = pointer1[w] + block_partition[w];
  return block_counter;
}

/* bmat_kernels.cuh:21-111.  hspa: T*128 floats, hind: T*8 ints, both fully written. */
void oracle_hmat_gen(const int32_t *node_pointer, const int32_t *edge_list, const int32_t *block_partition,
                     const int32_t *edge_to_column, const int32_t *edge_to_row, const int32_t *pointer1,
                     int32_t num_row_windows, int32_t num_nodes, float *hspa, int32_t *hind) {
  (void)block_partition;
  int64_t total = pointer1[num_row_windows];
  memset(hspa, 0, (size_t)total * BLK_H * BLK_W * sizeof(float)); /* :76-79 */
  memset(hind, 0, (size_t)total * BLK_W * sizeof(int32_t));       /* :71-73 */
  for (int32_t w = 0; w < num_row_windows; w++) {
    int32_t end_row = (w + 1) * BLK_H < num_nodes ? (w + 1) * BLK_H : num_nodes;
    for (int32_t e = node_pointer[w * BLK_H]; e < node_pointer[end_row]; e++) { /* :92-106 */
      int32_t col = edge_to_column[e];
      int64_t blk = (int64_t)pointer1[w] + col / BLK_W;
      int32_t row_local = edge_to_row[e] % BLK_H;
      int32_t col_local = col % BLK_W;
      hspa[blk * (BLK_H * BLK_W) + row_local * BLK_W + col_local] = 1.0f;
      hind[blk * BLK_W + col_local] = edge_list[e];
    }
  }
}

/* bmat_kernels.cuh:151-193 */
void oracle_hmat_packed_swizzle(int32_t num_row_windows, const int32_t *pointer1, const float *hspa,
                                uint32_t *hspa_packed) {
  int64_t total = pointer1[num_row_windows];
  for (int64_t b = 0; b < total; b++) {
    const float *tile = hspa + b * (BLK_H * BLK_W);
    for (int idx = 0; idx < 4; idx++) {
      uint32_t word = 0;
      for (int bit = 0; bit < 32; bit++) {
        int row = (bit >> 2) + 8 * (idx % 2); /* :180-183 */
        int col = (bit % 4) + 4 * (idx / 2);
        if (fabsf(tile[row * BLK_W + col] - 0.0f) > 1e-5f) word |= (1u << bit); /* :186-188 */
      }
      hspa_packed[b * 4 + idx] = word;
    }
  }
}

static inline float round_tf32_rna(float x) { /* cvt.rna.tf32.f32, spmm_kernels.cuh:1642 */
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7F800000u) != 0x7F800000u) u = (u + 0x1000u) & 0xFFFFE000u;
  memcpy(&x, &u, 4);
  return x;
}

/* fp32 -> IEEE binary16 (round-to-nearest-even, subnormals, overflow to inf) -> fp32, in integer
 * arithmetic so that it does not depend on compiler/ISA _Float16 support. */
static inline float round_fp16(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  uint32_t sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return x;                       /* inf / nan */
  if (a >= 0x477FF000u) { a = 0x7F800000u; }            /* >= 65520 rounds to inf */
  else if (a >= 0x38800000u) {                          /* normal half: keep 10 mantissa bits */
    uint32_t lsb = (a >> 13) & 1u;
    a = (a + 0xFFFu + lsb) & 0xFFFFE000u;
  } else if (a < 0x33000000u) { a = 0; }                /* < 2^-25 -> 0 (2^-25 itself ties to even = 0) */
  else {                                                /* subnormal half: quantum 2^-24 */
    float f;
    memcpy(&f, &a, 4);
    float q = f * 16777216.0f;                          /* exact scaling by 2^24 */
    float rq = nearbyintf(q);                           /* default rounding mode = RNE */
    f = rq * (1.0f / 16777216.0f);
    memcpy(&a, &f, 4);
  }
  a |= sign;
  memcpy(&x, &a, 4);
  return x;
}

/* rounding: 0 none, 1 tf32-rna (reference), 2 fp16 (gfx950 build).
 * spmm_kernels.cuh:1632-1716; computes the N%16 tail too (the oracle must). */
void oracle_spmm_blocked(const int32_t *pointer1, const uint32_t *hspa_packed, const int32_t *hind,
                         int32_t num_nodes, int32_t embedding_dim, const float *input, float *output,
                         int32_t rounding) {
  int32_t num_windows = (num_nodes + BLK_H - 1) / BLK_H;
  float *acc = (float *)malloc((size_t)BLK_H * embedding_dim * sizeof(float));
  float *brow = (float *)malloc((size_t)embedding_dim * sizeof(float));
  for (int32_t w = 0; w < num_windows; w++) {
    memset(acc, 0, (size_t)BLK_H * embedding_dim * sizeof(float)); /* :1598-1607 */
    for (int64_t b = pointer1[w]; b < pointer1[w + 1]; b++) {
      const uint32_t *words = hspa_packed + b * 4;
      for (int c = 0; c < BLK_W; c++) {
        /* column c lives in words t = 2*(c>>2) (rows 0-7) and t+1 (rows 8-15), bits 4*(r&7)+(c&3) */
        uint32_t lo = words[2 * (c >> 2)], hi = words[2 * (c >> 2) + 1];
        uint32_t colmask = 0x11111111u << (c & 3);
        if (((lo | hi) & colmask) == 0) continue; /* A==0 for all 16 rows: adds exact zeros */
        const float *src = input + (int64_t)hind[b * BLK_W + c] * embedding_dim;
        for (int f = 0; f < embedding_dim; f++)
          brow[f] = rounding == 1 ? round_tf32_rna(src[f]) : rounding == 2 ? round_fp16(src[f]) : src[f];
        for (int r = 0; r < BLK_H; r++) {
          uint32_t word = (r < 8) ? lo : hi;
          if ((word >> (4 * (r & 7) + (c & 3))) & 1u) {
            float *dst = acc + (size_t)r * embedding_dim;
            for (int f = 0; f < embedding_dim; f++) dst[f] += brow[f];
          }
        }
      }
    }
    for (int r = 0; r < BLK_H; r++) { /* :1687-1712 */
      int64_t row = (int64_t)w * BLK_H + r;
      if (row < num_nodes)
        memcpy(output + row * embedding_dim, acc + (size_t)r * embedding_dim,
               (size_t)embedding_dim * sizeof(float));
    }
  }
  free(acc);
  free(brow);
}

/* csr(ones) @ feat with fp64 accumulation; duplicates are summed like torch.sparse.mm. */
void oracle_spmm_csr(const int32_t *indptr, const int32_t *indices, int32_t num_nodes, int32_t embedding_dim,
                     const float *input, float *output, int32_t rounding) {
  double *acc = (double *)malloc((size_t)embedding_dim * sizeof(double));
  for (int32_t i = 0; i < num_nodes; i++) {
    for (int f = 0; f < embedding_dim; f++) acc[f] = 0.0;
    for (int32_t e = indptr[i]; e < indptr[i + 1]; e++) {
      const float *src = input + (int64_t)indices[e] * embedding_dim;
      for (int f = 0; f < embedding_dim; f++) {
        float v = rounding == 1 ? round_tf32_rna(src[f]) : rounding == 2 ? round_fp16(src[f]) : src[f];
        acc[f] += (double)v;
      }
    }
    for (int f = 0; f < embedding_dim; f++) output[(int64_t)i * embedding_dim + f] = (float)acc[f];
  }
  free(acc);
}
