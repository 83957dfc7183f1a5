// Voltrix-SpMM for MI355X (gfx950) -- stage-list executor: C-stationary SpMM over a precomputed schedule.
//
// Same math, same handle and the same per-stage machinery as spmm_tc16_kernel (spmm_kernels.hpp): a stage is up to 4
// consecutive TC blocks of one window, gathered by LDS-DMA into a wave-private ring and contracted with
// v_mfma_f32_16x16x32_f16.  What changes is WHO decides the order of the stages:
//
//   spmm_tc16_kernel   one wave = one (window, slab); it walks that window's blocks front to back.  Waves of an XCD
//                      that happen to run together share gathered rows through L2 only by accident of timing; on the
//                      headline graph 2/3 of the gathers miss L2 (17.5 GB of fabric traffic for 0.6 GB of compulsory
//                      bytes, profiles/r01).
//   spmm_list_kernel   one wave = G windows whose accumulators (G x FS/16 x 4 registers) stay resident while the wave
//                      executes a LIST of stages built ahead of time (schedule.hpp / voltrix/schedule.py).  The builder
//                      orders every wave's list the same way -- blocks near the diagonal window by window, then the far
//                      blocks panel by panel of B -- so all waves of an XCD sweep B's row panels together and each panel
//                      is fetched into that XCD's L2 about once per sweep.  C is written once (no partial sums in
//                      memory), results do not depend on timing (the order is data, not a race).
//
// Entry (int4): x = first TC block (global index), y = count (0..4 valid blocks; 0 = nothing to gather/multiply: empty
// window or list padding) | g << 8 (accumulator set) | flush << 16 (after this stage: store set g to window z, clear it),
// | tail << 17 (the stage may contain padded hind slots: a window's last block, a partial stage, count 0),
// z = window id, w = unused.  Every wave's list ends with 2*DEPTH+1 padding entries (count 0, a valid block) so that
// the pipeline issues exactly 1 + NDMA DMAs per step and one static s_waitcnt vmcnt immediate serves the whole list.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_kernels.hpp"

namespace voltrix {

typedef int int4v_t __attribute__((ext_vector_type(4)));  // plain vector: loadable from the constant address space

struct SpmmListArgs {
  const uint32_t* hspa_packed;
  const int* hind;
  const _Float16* input;
  float* output;
  const int4v_t* entries;  // all waves' lists, back to back
  const int* wave_ptr;     // [num_waves + 1] offsets into entries (the 2*DEPTH+1 padding entries included)
  int num_waves;
  int num_nodes;
  int F;
  int fs0;                 // first column of the slab this launch computes
};

template <int FS, int DEPTH, int WAVES, int G>
struct SpmmListTile {
  using Base = SpmmTile<FS, DEPTH, WAVES, 2>;
  static constexpr int GROUPS = G;
  static_assert(G >= 1 && G <= 8, "accumulator sets per wave");
};

template <class LT>
static __global__ __launch_bounds__(LT::Base::THREADS) void spmm_list_kernel(const SpmmListArgs a) {
  using T = typename LT::Base;
  constexpr int FS = T::FS, D = T::DEPTH, MS = T::META_SLOTS, G = LT::GROUPS;
  constexpr int ROW_BYTES = T::ROW_BYTES, STAGE_BYTES = T::STAGE_BYTES, NDMA = T::DMA_PER_STAGE;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int wid = blockIdx.x * T::WAVES + wave;
  if (wid >= a.num_waves) return;  // wave-uniform; no barriers in this kernel
  const int e_begin = a.wave_ptr[wid];
  const int n_all = a.wave_ptr[wid + 1] - e_begin;
  const int nreal = n_all - (2 * D + 1);  // the builder appends 2D+1 padding entries
  if (nreal <= 0) return;
  // constant address space: the list is read with scalar loads (SMEM, lgkmcnt).  A vector load here would put a
  // VGPR-destination op into the vmcnt stream and hipcc would drain the LDS-DMA pipeline with vmcnt(0) every step.
  using const_i4_ptr = const int4v_t __attribute__((address_space(4)))*;
  const const_i4_ptr ent = (const_i4_ptr)(a.entries + e_begin);
  const int F = a.F, fs0 = a.fs0;

  float4_t acc[G][SLOTS];
#pragma unroll
  for (int gi = 0; gi < G; ++gi)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[gi][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem + (unsigned)wave * T::WAVE_LDS;
  const unsigned meta0 = lds0 + D * STAGE_BYTES;

  // ---- lane constants (as in spmm_tc16_kernel) -----------------------------------------------------------------
  const int k32 = lane & 31, kblk = k32 >> 3, kcol = k32 & 7;
  const unsigned colmask = 0x11111111u << (kcol & 3);
  const unsigned vword_off = 128 + 4 * (4 * kblk + 2 * (kcol >> 2));
  const int g16 = lane >> 4, R = lane & 15;
  const unsigned a_shift = 4 * (R & 7);
  const int mj = (lane - 32) & 15;

  auto issue_meta = [&](const int4v_t e, int mslot) {
    const int cnt = e.y & 0xFF;
    const int last = e.x + (cnt > 0 ? cnt - 1 : 0);  // stay inside the stage's own blocks
    const void* src;
    if (lane < 32) {
      int blk = e.x + kblk;
      blk = blk < last ? blk : last;
      src = a.hind + (8ll * blk + kcol);
    } else {
      int blk = e.x + (mj >> 2);
      blk = blk < last ? blk : last;
      src = a.hspa_packed + (4ll * blk + (mj & 3));
    }
    dma_b32(src, meta0 + mslot * T::META_BYTES);
  };

  auto issue_data = [&](int dslot, const int (&hr)[NDMA]) {
    const unsigned dst = lds0 + dslot * STAGE_BYTES;
    auto piece = [&](auto kc, int ib) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDMA) {
        const int i = ib + K;
        const int r = i * RPD + lane / LPR;
        const int c = lane % LPR;
        int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r)) * 16) + (c & 1) * 8;
        col = col < F ? col : fs0;
        dma_b128_off<K * 1024>(a.input + ((long long)hr[i] * F + col), dst + ib * 1024);
      }
    };
#pragma unroll
    for (int ib = 0; ib < NDMA; ib += 4) {
      piece(std::integral_constant<int, 0>{}, ib);
      piece(std::integral_constant<int, 1>{}, ib);
      piece(std::integral_constant<int, 2>{}, ib);
      piece(std::integral_constant<int, 3>{}, ib);
    }
  };
  auto rows_from_columns = [&](int hv, int (&hr)[NDMA]) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) hr[i] = __shfl(hv, i * RPD + lane / LPR, kWave);
  };
  const unsigned hr_off = 4 * (lane / LPR);
  auto read_rows = [&](unsigned mbase, int (&hr)[NDMA]) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) hr[i] = (int)lds_read_b32(mbase + hr_off + 4 * (i * RPD));
  };

  // hraw: this lane's hind word; h_first: hind word 0 of the stage (first column of its first block: a real column of
  // the window whenever count > 0; for count == 0 entries it only has to be a valid row, and the builder guarantees it)
  auto sanitise = [&](const int4v_t e, unsigned hraw, uint2_t vw, int h_first) -> int {
    const int cnt = e.y & 0xFF;
    const bool valid = (kblk < cnt) && (((vw[0] | vw[1]) & colmask) != 0u);
    return valid ? (int)hraw : h_first;
  };

  auto flush = [&](float4_t (&set)[SLOTS], int w) {
    const int orow0 = w * kBlkH + 4 * (lane >> 4);
    const int ocol0 = fs0 + (lane & 15);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int col = ocol0 + 16 * s;
      if (col < F) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = orow0 + j;
          if (row < a.num_nodes) a.output[(long long)row * F + col] = set[s][j] * kAScaleInv;
        }
      }
      set[s] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < D; ++j) issue_meta(ent[j], j);
  wait_vmcnt<0>();
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue_meta(ent[D + j], (D + j) % MS);
    const unsigned m = meta0 + j * T::META_BYTES;
    const unsigned hraw = lds_read_b32(m + 4 * k32);
    const unsigned hfirst = lds_read_b32(m);
    const uint2_t vw = lds_read_b64(m + vword_off);
    wait_lgkmcnt0();
    int hr[NDMA];
    rows_from_columns(sanitise(ent[j], hraw, vw, (int)hfirst), hr);
    issue_data(j, hr);
  }

  // ---- main loop: every step retires one stage and issues exactly 1 + NDMA DMAs ------------------------------------
  int dslot = 0, mslot = 0, mslot_d = D, mslot_2d = (2 * D) % MS;
  int4v_t e_c = ent[0], e_d = ent[D], e_m = ent[2 * D];
  for (int t = 0; t < nreal; ++t) {
    wait_vmcnt<T::vm_behind(D - 1)>();

    const unsigned mt = meta0 + mslot * T::META_BYTES;
    const unsigned md = meta0 + mslot_d * T::META_BYTES;
    // bit 17 of the entry: the stage may hold padded / masked columns (a window's last block, a partial stage, count 0)
    const bool tail_stage = (e_d.y >> 17) & 1;  // wave-uniform
    unsigned hraw = 0, hfirst = 0;
    uint2_t vw = {0u, 0u};
    int hr[NDMA];
    if (tail_stage) {
      hraw = lds_read_b32(md + 4 * k32);
      hfirst = lds_read_b32(md);
      vw = lds_read_b64(md + vword_off);
    } else {
      read_rows(md, hr);
    }
    const unsigned wlo = lds_read_b32(mt + 128 + 4 * (4 * g16 + (R >> 3)));
    const unsigned whi = lds_read_b32(mt + 128 + 4 * (4 * g16 + 2 + (R >> 3)));
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int trow = 8 * g16 + q;
    const unsigned dbase = lds0 + dslot * STAGE_BYTES + trow * ROW_BYTES + 8 * p;
    const int tr_z = slot_swizzle<SLOTS>(trow);
    uint2_t blo[SLOTS], bhi[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const unsigned addr = dbase + ((s ^ tr_z) << 5);
      blo[s] = lds_read_tr16_b64<0>(addr);
      bhi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(addr);
    }
    wait_lgkmcnt0();

    // next step's entries: scalar loads issued HERE, right after the only lgkmcnt(0) of the step, so that their
    // latency hides under the DMA issue + MFMA phase (SMEM returns out of order: any lgkmcnt(0) drains it)
    const int nt1 = t + 1;
    const int4v_t n_c = ent[nt1], n_d = ent[nt1 + D];
    const int4v_t n_m = (nt1 + 2 * D) < n_all ? ent[nt1 + 2 * D] : ent[n_all - 1];
    __builtin_amdgcn_sched_barrier(0);

    issue_meta(e_m, mslot_2d);
    if (tail_stage) rows_from_columns(sanitise(e_d, hraw, vw, (int)hfirst), hr);
    issue_data(dslot, hr);

    const int cnt = e_c.y & 0xFF, gi = (e_c.y >> 8) & 0xFF;
    if (cnt > 0) {  // count 0: empty window or padding -- its LDS slot holds rows that must not reach the matrix core
      unsigned nl = (wlo >> a_shift) & 0xFu, nh = (whi >> a_shift) & 0xFu;
      if (g16 >= cnt) nl = nh = 0u;
      const half8_t afrag = nibbles_to_half8_x2(nl, nh);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        if (gi == k) {  // wave-uniform: accumulator sets are indexed statically
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) {
            const uint4_t bq = {blo[s][0], blo[s][1], bhi[s][0], bhi[s][1]};
            acc[k][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq), acc[k][s], 0, 0, 0);
          }
        }
      }
    }
    if ((e_c.y >> 16) & 1) {
#pragma unroll
      for (int k = 0; k < G; ++k)
        if (gi == k) flush(acc[k], e_c.z);
    }

    e_c = n_c;
    e_d = n_d;
    e_m = n_m;
    dslot = dslot + 1 == D ? 0 : dslot + 1;
    mslot = mslot + 1 == MS ? 0 : mslot + 1;
    mslot_d = mslot_d + 1 == MS ? 0 : mslot_d + 1;
    mslot_2d = mslot_2d + 1 == MS ? 0 : mslot_2d + 1;
  }
  wait_vmcnt<0>();
}

template <class LT>
inline int launch_spmm_list(const uint32_t* hspa_packed, const int* hind, int num_nodes, int embedding_dim,
                            const _Float16* input, float* output, const int4v_t* entries, const int* wave_ptr,
                            int num_waves, hipStream_t stream) {
  using T = typename LT::Base;
  if (num_nodes < 0 || embedding_dim < 0 || num_waves < 0) return kErrBadShape;
  if (num_nodes == 0 || embedding_dim == 0 || num_waves == 0) return kOk;
  if (embedding_dim % 8 != 0) return kErrBadShape;
  if (((uintptr_t)input & 15) || ((uintptr_t)hspa_packed & 15) || ((uintptr_t)entries & 15)) return kErrBadShape;
  const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_list_kernel<LT>), T::BLOCK_LDS);
  if (lds_rc != kOk) return lds_rc;
  SpmmListArgs a;
  a.hspa_packed = hspa_packed;
  a.hind = hind;
  a.input = input;
  a.output = output;
  a.entries = entries;
  a.wave_ptr = wave_ptr;
  a.num_waves = num_waves;
  a.num_nodes = num_nodes;
  a.F = embedding_dim;
  const int grid = (num_waves + T::WAVES - 1) / T::WAVES;
  const int num_slabs = (embedding_dim + T::FS - 1) / T::FS;
  for (int s = 0; s < num_slabs; ++s) {  // one launch per FS-column slab (F = 128 with FS = 128: one launch)
    a.fs0 = s * T::FS;
    hipLaunchKernelGGL(spmm_list_kernel<LT>, dim3(grid), dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  }
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
