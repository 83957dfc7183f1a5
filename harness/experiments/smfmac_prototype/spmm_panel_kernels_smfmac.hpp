// Voltrix-SpMM for MI355X (gfx950) -- panel kernel: the "shared column" half of the two-level condensed format.
//
// The reference format (and spmm_tc16_kernel) condenses columns per 16-row window: a gathered row of B serves 16 rows
// of A and, on graphs with a few hundred edges per row, about ONE of them (TC-block fill 6-7 %): every edge costs one
// 2*F-byte row gather out of L2 / Infinity Cache, and that gather traffic -- not HBM, not the matrix cores -- is what
// bounds the kernel.  Columns that are referenced by SEVERAL rows of a taller row panel (community /
// band structure, hub columns) can do better: gathered once per panel into LDS, shared by all the panel's windows.
//
//   panel       PANEL_ROWS = WAVES * RB * 16 consecutive rows (256 or 512); one workgroup per (panel, feature slab)
//   plan        per panel the sorted list of its shared columns (those with >= tau edges inside the panel; chosen by
//               the plan builder, panel_plan.hpp), cut into k-steps of 64 columns:
//                 panel_ptr  int32 [NP+1]            first k-step of every panel
//                 panel_cols int32 [64 * (S + 1)]    row of B per (k-step, position); unused slots repeat a real column
//                 panel_bits uint32 [(S + 1) * WAVES * 64 * 4]  four words per (k-step, wave v, lane L = 16 g + R):
//                            a kept edge (row 16 (RB v + j) + R of the panel, position 16 g + 4 q + pos of the k-step,
//                            t-th kept edge of that row in its group of four positions, t < 2) sets bit 16 t + 4 j + q
//                            of word 0 and stores pos in bits [16 (j & 1) + 2 (2 q + t), +2) of word 1 + (j >> 1)
//   2:4 rule    a row keeps at most two edges per group of four consecutive positions (the builder moves the rare third
//               to the residual), so the adjacency is a 2:4 structured-sparse A operand: v_smfmac_f32_16x16x64_f16 takes
//               64 columns per instruction and issues 1.75x faster than the dense 16x16x32 (measured:
//               harness/experiments/smfmac_rate.py) -- 3.5x the columns per second at a panel density of ~3 %.
//   everything else (columns below tau, demoted edges) stays in the reference's window format and runs through
//   spmm_tc16_kernel; the two results are added (two addends per element: the sum does not depend on order).
//
// Per k-step the workgroup gathers 64 rows of B ONCE (16 KiB at FS = 128: two 32-row images; LDS-DMA, every wave issues its
// share) and each wave multiplies it into RB 16-row blocks: RB * FS/16 smfmac per 4 * FS/16 transposed LDS reads.  The
// ring is shared, so there is one raw s_barrier per step: counted vmcnt wait -> barrier -> reads (cdna_hip_programming.md
// "Pipelining across barriers"); this wave's 1 KiB of adjacency words travels with the step's rows, the step's 64
// column ids a ring ahead (both wave-private LDS-DMAs).
//
// Operand pairing of v_smfmac_f32_16x16x64_f16 (decoded with harness/experiments/smfmac_probe.py): A lane 16 g + R holds
// row R's kept values for positions 16 g .. 16 g + 15 of the step (register q = the two kept values of group q, their
// 2-bit positions at bits [4 q, 4 q + 4) of the 16-bit index half selected by ABID); B lane 16 b + n holds column n of rows
// 8 b .. 8 b + 7 of the first 32-row image in elements 0-7 and of the second image in elements 8-15.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_kernels.hpp"

// Diagnostic builds only (harness/experiments/panel_diag.py): bit 0 skips the MFMAs, bit 1 the row DMAs, bit 2 the
// barrier.  Results are wrong by design; shipped kernels use 0.
#ifndef VOLTRIX_PANEL_DIAG
#define VOLTRIX_PANEL_DIAG 0
#endif

namespace voltrix {

typedef _Float16 half16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x16_t __attribute__((ext_vector_type(16)));
typedef unsigned uint8v_t __attribute__((ext_vector_type(8)));

constexpr int kPanelK = 64;   // shared columns per k-step

//   FS     feature slab per workgroup (columns of B / C): 32, 64 or 128
//   DEPTH  ring slots (k-steps of gathered rows in flight per workgroup), >= 2
//   WAVES  waves per workgroup (4 or 8)
//   RB     16-row blocks per wave (2 or 4): PANEL_ROWS = WAVES * RB * 16
template <int FS_, int DEPTH_, int WAVES_, int RB_, bool BF16_ = false>
struct PanelTile {
  static constexpr int FS = FS_, DEPTH = DEPTH_, WAVES = WAVES_, RB = RB_;
  static constexpr bool BF16 = BF16_;
  static_assert(FS == 32 || FS == 64 || FS == 128, "feature slab");
  static_assert(RB >= 1 && RB <= 4, "a lane's adjacency words hold four row blocks");
  static_assert(DEPTH >= 2 && DEPTH <= 8, "ring depth");
  static constexpr int PANEL_ROWS = WAVES * RB * 16;
  static constexpr int THREADS = WAVES * kWave;
  static constexpr int ROW_BYTES = FS * 2;
  static constexpr int IMAGE_BYTES = 32 * ROW_BYTES;               // 32 gathered rows (the unit of the LDS swizzle)
  static constexpr int STAGE_BYTES = 2 * IMAGE_BYTES;              // one k-step
  static constexpr int NDMA = STAGE_BYTES / 1024;                  // 1 KiB per global_load_lds_dwordx4
  // every wave issues the same number of row DMAs (static vmcnt); with more waves than DMAs the surplus waves repeat
  // the first ones (same bytes to the same place: harmless, and only at FS = 32 where a step is 4 KiB)
  static_assert(NDMA % WAVES == 0 || WAVES % NDMA == 0, "row DMAs per step vs waves");
  static constexpr int DPW = NDMA >= WAVES ? NDMA / WAVES : 1;     // row DMAs per wave and step
  static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;
  static constexpr int LANES_PER_ROW = ROW_BYTES / 16;
  static constexpr int SLOTS = FS / 16;
  // wave-private rings: adjacency words (1 KiB per step, travel with the rows) and column ids (256 B, DEPTH-1 steps ahead)
  static constexpr int BITS_BYTES = 1024;
  static constexpr int COLS_BYTES = 256;
  static constexpr int COLS_SLOTS = 2 * DEPTH - 1;
  static constexpr int WAVE_META = DEPTH * BITS_BYTES + COLS_SLOTS * COLS_BYTES;
  static constexpr int VM_PER_STEP = DPW + 2;                      // LDS-DMAs per wave and step
  static constexpr int DATA_LDS = DEPTH * STAGE_BYTES;
  static constexpr int BLOCK_LDS = DATA_LDS + WAVES * WAVE_META;
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
  static_assert(VM_PER_STEP * (DEPTH - 2) <= 63, "vmcnt is a 6-bit counter on gfx9");
};

// Adjacency word -> structured-sparse A fragment.  Bit p (p = 4 j + q) of the word says that the first kept value of
// group q of row block j is present, bit 16 + p the second, so one shift + one mask yields the packed fp16 pair
// {2.0 or 0.0} x 2 of register q (2.0 = 0x4000; the 0.5 is applied once in the epilogue, as in spmm_tc16_kernel).
__device__ __forceinline__ half8_t adjacency_to_half8_x2(unsigned w, int p0) {  // p0 = 4 j: constant after unrolling
  uint4_t r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + i;
    r[i] = (p <= 14 ? (w << (14 - p)) : (w >> (p - 14))) & 0x40004000u;
  }
  return __builtin_bit_cast(half8_t, r);
}

template <class T>
struct PanelArgs {
  using in_t = typename std::conditional<T::BF16, bfloat16_bits, _Float16>::type;
  const int* panel_ptr;        // [NP+1]
  const int* panel_cols;       // [64 * (S + 1)]
  const uint32_t* panel_bits;  // [(S + 1) * WAVES * 64 * 4]
  const int* panel_order;      // optional: launch position -> panel (longest first); nullptr = natural
  const in_t* input;
  float* output;
  const float* out_scale;      // optional device scalar (see SpmmArgs::out_scale)
  int num_nodes;
  int num_panels;
  int panels_per_xcd;
  int F;
  int accumulate;              // 1: C += A_shared * B (C holds the window kernel's part); 0: C = A_shared * B
};

template <class T>
static __global__ __launch_bounds__(T::THREADS) void spmm_panel_kernel(const PanelArgs<T> a) {
  constexpr int FS = T::FS, D = T::DEPTH, CS = T::COLS_SLOTS, RB = T::RB;
  constexpr int ROW_BYTES = T::ROW_BYTES, STAGE_BYTES = T::STAGE_BYTES, DPW = T::DPW;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));

  // XCD x = blockIdx.x % 8 owns a contiguous range of launch positions: neighbouring panels share most of their
  // columns (band / community structure), so they should share an L2.
  const int xcd = blockIdx.x % kNumXcd;
  const int pos = xcd * a.panels_per_xcd + (int)(blockIdx.x / kNumXcd);
  const int pos_end = (xcd + 1) * a.panels_per_xcd < a.num_panels ? (xcd + 1) * a.panels_per_xcd : a.num_panels;
  if (pos >= pos_end) return;  // workgroup-uniform
  const int panel = a.panel_order ? a.panel_order[pos] : pos;
  const int fs0 = blockIdx.y * FS;
  const int F = a.F;

  const int ks0 = a.panel_ptr[panel];
  const int nks = a.panel_ptr[panel + 1] - ks0;
  if (nks == 0 && a.accumulate) return;  // workgroup-uniform: nothing to add

  float4_t acc[RB][SLOTS];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[j][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned data0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned bits0 = data0 + T::DATA_LDS + (unsigned)wave * T::WAVE_META;
  const unsigned cols0 = bits0 + D * T::BITS_BYTES;

  if (nks > 0) {
    // ---- lane constants ---------------------------------------------------------------------------------------
    const unsigned row_bytes = (unsigned)F * 2u;
    const int dma0 = (wave * DPW) % T::NDMA;      // this wave's first row DMA of a step: LDS bytes [dma0 KiB, ...)
    const char* cbase[DPW];   // source of this lane's 16 bytes in row DMA d of a step, before the row offset
    unsigned hr_off[DPW];     // byte offset of that DMA's row id inside the step's column list
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
      const int r = (dma0 + d) * RPD + lane / LPR;  // gathered row inside the step (0 .. 63)
      const int c = lane % LPR;                     // 16-byte chunk inside the row
      int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r & 31)) * 16) + (c & 1) * 8;  // swizzle on the SOURCE
      col = col < F ? col : fs0;                    // F % FS tail: stay in bounds, never stored
      unsigned long long cb = (unsigned long long)((const char*)a.input + (long long)col * 2);
      asm volatile("" : "+v"(cb));
      cbase[d] = (const char*)cb;
      hr_off[d] = 4 * r;
    }
    // steps past the panel's end repeat its last step (the pipeline issues a static number of DMAs)
    const uint32_t* const bits_base = a.panel_bits + (((long long)ks0 * T::WAVES + wave) * kWave + lane) * 4;
    const int* const cols_base = a.panel_cols + (long long)ks0 * kPanelK + lane;
    auto issue_cols = [&](int s) {
      const int sc = s < nks ? s : nks - 1;
      dma_b32(cols_base + (long long)sc * kPanelK, cols0 + (unsigned)(s % CS) * T::COLS_BYTES);
    };
    auto issue_rows_bits = [&](int s) {
      const int sc = s < nks ? s : nks - 1;
      const unsigned cslot = cols0 + (unsigned)(s % CS) * T::COLS_BYTES;
      const unsigned dst = data0 + (unsigned)(s % D) * STAGE_BYTES + (unsigned)dma0 * 1024u;
      unsigned hrow[DPW];
#pragma unroll
      for (int d = 0; d < DPW; ++d) hrow[d] = lds_read_b32(cslot + hr_off[d]);
      wait_lgkmcnt0();
#pragma unroll
      for (int d = 0; d < DPW; ++d)
        if (!(VOLTRIX_PANEL_DIAG & 2)) dma_b128(cbase[d] + (unsigned long long)hrow[d] * row_bytes, dst + d * 1024);
      dma_b128(bits_base + (long long)sc * (T::WAVES * kWave * 4), bits0 + (unsigned)(s % D) * T::BITS_BYTES);
    };

    // ---- prologue: column ids of steps 0 .. D-2, then the virtual steps -(D-1) .. -1 -----------------------------
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue_cols(s);
    wait_vmcnt<0>();
#pragma unroll
    for (int s = 0; s < D - 1; ++s) {
      issue_rows_bits(s);
      issue_cols(s + D - 1);
    }

    // MFMA lane roles: B column R, rows 8g+q (+4) of either image (as in spmm_tc16_kernel)
    const int g = lane >> 4;
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int trow = 8 * g + q;
    const unsigned rd_off = trow * ROW_BYTES + 8 * p;
    const int tr_z = slot_swizzle<SLOTS>(trow);

    for (int t = 0; t < nks; ++t) {
      // rows + adjacency words of step t (issued D-1 steps ago) and the column ids of step t+D-1 must have landed; the
      // D-2 younger steps may stay in flight.  Steps past nks-D+1 issue nothing.
      const int young = nks - 1 - t;
      if (young >= D - 2) {
        wait_vmcnt<T::VM_PER_STEP*(D - 2)>();
      } else {
        switch (young) {
          case 0: wait_vmcnt<0>(); break;
          case 1: wait_vmcnt<T::VM_PER_STEP * 1>(); break;
          case 2: wait_vmcnt<T::VM_PER_STEP * 2>(); break;
          case 3: wait_vmcnt<T::VM_PER_STEP * 3>(); break;
          case 4: wait_vmcnt<T::VM_PER_STEP * 4>(); break;
          default: wait_vmcnt<T::VM_PER_STEP * 5>(); break;
        }
      }
      if (!(VOLTRIX_PANEL_DIAG & 4))
        __builtin_amdgcn_s_barrier();  // every wave's share of step t has landed; everyone is done reading step t-1
      __builtin_amdgcn_sched_barrier(0);

      if (t + D - 1 < nks) {           // workgroup-uniform
        issue_rows_bits(t + D - 1);    // into the slot step t-1 has just left
        issue_cols(t + 2 * D - 2);
      }

      const uint4_t aw = lds_read_b128(bits0 + (unsigned)(t % D) * T::BITS_BYTES + 16 * lane);
      const unsigned dt = data0 + (unsigned)(t % D) * STAGE_BYTES + rd_off;
      uint2_t b0lo[SLOTS], b0hi[SLOTS], b1lo[SLOTS], b1hi[SLOTS];
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const unsigned addr = dt + ((s ^ tr_z) << 5);
        b0lo[s] = lds_read_tr16_b64<0>(addr);
        b0hi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(addr);
        b1lo[s] = lds_read_tr16_b64<T::IMAGE_BYTES>(addr);
        b1hi[s] = lds_read_tr16_b64<T::IMAGE_BYTES + 4 * ROW_BYTES>(addr);
      }
      wait_lgkmcnt0();
      // The LDS reads above are asynchronous inline asm: a register of theirs that nothing reads afterwards (word 3 of
      // the adjacency words is always 0) would be handed out again while the read is still in flight, and the late
      // write-back would land in its new owner.  Keep the whole tuple alive across the wait.
      asm volatile("" ::"v"(aw));
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const half8_t afrag = adjacency_to_half8_x2(aw[0], 4 * j);
        const int idx = (int)aw[1 + (j >> 1)];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          const uint8v_t bq = {b0lo[s][0], b0lo[s][1], b0hi[s][0], b0hi[s][1], b1lo[s][0], b1lo[s][1], b1hi[s][0], b1hi[s][1]};
          if (VOLTRIX_PANEL_DIAG & 1) {
            asm volatile("" ::"v"(afrag), "v"(bq));
            continue;
          }
          if constexpr (T::BF16) {
            if (j & 1)
              acc[j][s] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                    __builtin_bit_cast(bf16x16_t, bq), acc[j][s], idx, 0, 1);
            else
              acc[j][s] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                    __builtin_bit_cast(bf16x16_t, bq), acc[j][s], idx, 0, 0);
          } else {
            if (j & 1)
              acc[j][s] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(afrag, __builtin_bit_cast(half16_t, bq), acc[j][s], idx,
                                                                   0, 1);
            else
              acc[j][s] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(afrag, __builtin_bit_cast(half16_t, bq), acc[j][s], idx,
                                                                   0, 0);
          }
        }
      }
    }
    wait_vmcnt<0>();  // nothing of this workgroup may still be writing LDS when it is released
  }

  // ---- epilogue: D[row = 4*(lane>>4) + i][col = lane & 15] per (row block, 16-column slot) ------------------------
  const float oscale = kAScaleInv * (a.out_scale ? *a.out_scale : 1.0f);
  const int prow0 = panel * T::PANEL_ROWS + wave * (RB * 16) + 4 * (lane >> 4);
  const int ocol0 = fs0 + (lane & 15);
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    // accumulate mode: all of a row block's loads first (one round trip, not one per element), then add and store
    float prev[SLOTS][4];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = prow0 + 16 * j + i, col = ocol0 + 16 * s;
        prev[s][i] = (a.accumulate && col < F && row < a.num_nodes) ? a.output[(long long)row * F + col] : 0.0f;
      }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = prow0 + 16 * j + i, col = ocol0 + 16 * s;
        if (col < F && row < a.num_nodes) a.output[(long long)row * F + col] = prev[s][i] + acc[j][s][i] * oscale;
      }
    }
  }
}

// Host launcher.  The plan arrays must be padded as the builder pads them: one k-step beyond S = panel_ptr[NP].
template <class T>
inline int launch_spmm_panel(const int* panel_ptr, const int* panel_cols, const uint32_t* panel_bits,
                             const int* panel_order, int num_nodes, int embedding_dim, const void* input, float* output,
                             int accumulate, const float* out_scale, hipStream_t stream) {
  if (num_nodes < 0 || embedding_dim < 0) return kErrBadShape;
  if (num_nodes == 0 || embedding_dim == 0) return kOk;
  if (embedding_dim % 8 != 0 || ((uintptr_t)input & 15)) return kErrBadShape;
  PanelArgs<T> a;
  a.panel_ptr = panel_ptr;
  a.panel_cols = panel_cols;
  a.panel_bits = panel_bits;
  a.panel_order = panel_order;
  a.input = static_cast<const typename PanelArgs<T>::in_t*>(input);
  a.output = output;
  a.out_scale = out_scale;
  a.num_nodes = num_nodes;
  a.num_panels = (num_nodes + T::PANEL_ROWS - 1) / T::PANEL_ROWS;
  a.panels_per_xcd = (a.num_panels + kNumXcd - 1) / kNumXcd;
  a.F = embedding_dim;
  a.accumulate = accumulate;
  const int slabs = (embedding_dim + T::FS - 1) / T::FS;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    if (T::BLOCK_LDS > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_panel_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, T::BLOCK_LDS) != hipSuccess)
      return kErrBadConfig;
    attr_done = true;
  }
  hipLaunchKernelGGL(spmm_panel_kernel<T>, dim3((unsigned)(a.panels_per_xcd * kNumXcd), (unsigned)slabs),
                     dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// dst += src (float32, count % 4 == 0, 16-byte aligned): joins the two halves of the two-level format when the window
// kernel and the panel kernel ran side by side on two streams into two buffers.
static __global__ __launch_bounds__(256) void add_inplace_f32_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                                const long long n4) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 d = reinterpret_cast<float4*>(dst)[i];
    const float4 x = reinterpret_cast<const float4*>(src)[i];
    d.x += x.x;
    d.y += x.y;
    d.z += x.z;
    d.w += x.w;
    reinterpret_cast<float4*>(dst)[i] = d;
  }
}

inline int add_inplace_f32(float* dst, const float* src, long long count, hipStream_t stream) {
  if (count < 0 || (count % 4) != 0 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return kErrBadShape;
  if (count == 0) return kOk;
  const long long n4 = count / 4;
  const int blocks = (int)(n4 / 256 + 1 < 256 * 16 ? n4 / 256 + 1 : 256 * 16);
  hipLaunchKernelGGL(add_inplace_f32_kernel, dim3(blocks), dim3(256), 0, stream, dst, src, n4);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
