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
//               the plan builder, voltrix/hybrid.py), cut into k-steps of 32 columns:
//                 panel_ptr  int32 [NP+1]            first k-step of every panel
//                 panel_cols int32 [32 * (S + pad)]  row of B per (k-step, k); unused slots repeat a real column
//                 panel_bits uint32 [(S + 1) * WAVES * 64] adjacency bits in MFMA A-operand order: word (k-step, wave v,
//                                                    lane L = 16 g + R), byte j, bit c  <=>  edge (row 16 (RB v + j) + R
//                                                    of the panel, column 8 g + c of the k-step)
//   everything else (columns below tau) stays in the reference's window format and runs through spmm_tc16_kernel; this
//   kernel then adds its share onto C (accumulate = 1) -- two addends per element, so the sum does not depend on order.
//
// Per k-step the workgroup gathers 32 rows of B ONCE (8 KiB at FS = 128; LDS-DMA, every wave issues its share) and each
// wave multiplies it into RB 16-row blocks: RB * FS/16 v_mfma_f32_16x16x32_f16 per 2 * FS/16 transposed LDS reads.  The
// ring is shared, so there is one raw s_barrier per step: counted vmcnt wait -> barrier -> reads (cdna_hip_programming.md
// "Pipelining across barriers"); the metadata (this wave's 256 B of adjacency bits, the step's 32 rows) is fetched by
// wave-private LDS-DMAs a ring ahead, as in spmm_tc16_kernel.
//
// Bound: matrix cores (16 rows x 32 columns per MFMA at the panel's density), with 1/16 .. 1/32 of the window kernel's
// gather traffic per covered edge.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_kernels.hpp"

// Diagnostic builds only (harness/experiments/panel_diag.py): bit 0 skips the MFMAs, bit 1 the row DMAs, bit 2 the
// barrier, bit 3 the adjacency expansion, bit 4 the fragment reads.  Results are wrong by design; shipped kernels use 0.
#ifndef VOLTRIX_PANEL_DIAG
#define VOLTRIX_PANEL_DIAG 0
#endif

namespace voltrix {

typedef int panel_int4_t __attribute__((ext_vector_type(4)));  // plain vector: loadable from the constant address space

//   FS     feature slab per workgroup (columns of B / C): 32, 64 or 128
//   DEPTH  ring slots (k-step groups of gathered rows in flight per workgroup)
//   WAVES  waves per workgroup (4 or 8)
//   RB     16-row blocks per wave (2 or 4): PANEL_ROWS = WAVES * RB * 16
//   KS     k-steps (of 32 columns) per ring slot / barrier
template <int FS_, int DEPTH_, int WAVES_, int RB_, int KS_ = 1, bool BF16_ = false>
struct PanelTile {
  static constexpr int FS = FS_, DEPTH = DEPTH_, WAVES = WAVES_, RB = RB_, KS = KS_;
  static constexpr bool BF16 = BF16_;
  static_assert(FS == 32 || FS == 64 || FS == 128, "feature slab");
  static_assert(RB >= 1 && RB <= 4, "a lane's adjacency word holds four row blocks");
  static_assert(KS == 1 || KS == 2, "one index DMA covers 64 columns");
  static_assert(DEPTH >= 3 && DEPTH <= 8, "ring depth");
  static constexpr int PANEL_ROWS = WAVES * RB * 16;
  static constexpr int THREADS = WAVES * kWave;
  static constexpr int ROW_BYTES = FS * 2;
  static constexpr int KSTEP_BYTES = kStageK * ROW_BYTES;          // 32 gathered rows
  static constexpr int STAGE_BYTES = KS * KSTEP_BYTES;
  static constexpr int NDMA = STAGE_BYTES / 1024;                  // 1 KiB per global_load_lds_dwordx4
  // every wave issues the same number of row DMAs (static vmcnt); with more waves than DMAs the surplus waves repeat
  // the first ones (same bytes to the same place: harmless, and only at FS = 32 where a step is 2-4 KiB)
  static_assert(NDMA % WAVES == 0 || WAVES % NDMA == 0, "row DMAs per step vs waves");
  static constexpr int DPW = NDMA >= WAVES ? NDMA / WAVES : 1;     // row DMAs per wave and step
  static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;
  static constexpr int LANES_PER_ROW = ROW_BYTES / 16;
  static constexpr int SLOTS = FS / 16;
  static constexpr int CPW = DPW * ROWS_PER_DMA;                   // consecutive column ids a wave gathers per step
  // wave-private ring of adjacency words: KS x 64 words per step, DEPTH steps (they travel with the step's rows)
  static constexpr int BITS_BYTES = KS * 256;
  static constexpr int VM_PER_STEP = DPW + KS;                     // LDS-DMAs per wave and step
  static constexpr int DATA_LDS = DEPTH * STAGE_BYTES;
  static constexpr int BLOCK_LDS = DATA_LDS + WAVES * DEPTH * BITS_BYTES;
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
  static_assert(VM_PER_STEP * (DEPTH - 2) <= 63, "vmcnt is a 6-bit counter on gfx9");
  static_assert(CPW % 4 == 0 && CPW <= 16, "column ids are fetched with s_load_dwordx4/x8/x16");
};

template <class T>
struct PanelArgs {
  using in_t = typename std::conditional<T::BF16, bfloat16_bits, _Float16>::type;
  const int* panel_ptr;        // [NP+1]
  const int* panel_cols;       // [32 * (S + 2)]
  const uint32_t* panel_bits;  // [(S + 1) * WAVES * 64]
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
  constexpr int FS = T::FS, D = T::DEPTH, KS = T::KS, RB = T::RB;
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
  const int ngroups = (nks + KS - 1) / KS;
  if (ngroups == 0 && a.accumulate) return;  // workgroup-uniform: nothing to add

  float4_t acc[RB][SLOTS];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[j][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned data0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned bits0 = data0 + T::DATA_LDS + (unsigned)wave * (D * T::BITS_BYTES);

  if (ngroups > 0) {
    // ---- lane constants ---------------------------------------------------------------------------------------
    const unsigned row_bytes = (unsigned)F * 2u;
    const int dma0 = (wave * DPW) % T::NDMA;      // this wave's first row DMA of a step: LDS bytes [dma0 KiB, ...)
    const int lq = lane / LPR;                    // row of a DMA this lane copies
    const char* cbase[DPW];                       // source of this lane's 16 bytes in row DMA d, before the row offset
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
      const int r = (dma0 + d) * RPD + lq;        // gathered row inside the step (0 .. 32 KS - 1)
      const int c = lane % LPR;                   // 16-byte chunk inside the row
      int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r & 31)) * 16) + (c & 1) * 8;  // swizzle on the SOURCE
      col = col < F ? col : fs0;                  // F % FS tail: stay in bounds, never stored
      unsigned long long cb = (unsigned long long)((const char*)a.input + (long long)col * 2);
      asm volatile("" : "+v"(cb));
      cbase[d] = (const char*)cb;
    }
    // Column ids: the CPW rows this wave gathers per step are wave-uniform -> scalar loads (constant address space:
    // SMEM counts on lgkmcnt, so nothing with a VGPR destination enters the vmcnt stream of the LDS-DMAs).
    using const_i4_ptr = const panel_int4_t __attribute__((address_space(4)))*;
    const const_i4_ptr cols4 = (const_i4_ptr)(a.panel_cols + ((long long)ks0 * kStageK + dma0 * RPD));
    struct Cols { panel_int4_t v[T::CPW / 4]; };
    auto load_cols = [&](int s) {                 // ids of step s (k-step group s of the panel)
      Cols c;
#pragma unroll
      for (int i = 0; i < T::CPW / 4; ++i) c.v[i] = cols4[(long long)s * (KS * kStageK / 4) + i];
      return c;
    };
    const uint32_t* const bits_base = a.panel_bits + ((long long)ks0 * T::WAVES + wave) * kWave + lane;
    // rows + adjacency words of step s: DPW + KS LDS-DMAs
    auto issue_step = [&](int s, const Cols& c) {
      const unsigned dst = data0 + (unsigned)(s % D) * STAGE_BYTES + (unsigned)dma0 * 1024u;
#pragma unroll
      for (int d = 0; d < DPW; ++d) {
        int hrow = c.v[(d * RPD) / 4][(d * RPD) % 4];
#pragma unroll
        for (int qq = 1; qq < RPD; ++qq) hrow = (lq == qq) ? c.v[(d * RPD + qq) / 4][(d * RPD + qq) % 4] : hrow;
        if (!(VOLTRIX_PANEL_DIAG & 2)) dma_b128(cbase[d] + (unsigned long long)(unsigned)hrow * row_bytes, dst + d * 1024);
        else asm volatile("" ::"v"(hrow));
      }
      const unsigned bdst = bits0 + (unsigned)(s % D) * T::BITS_BYTES;
#pragma unroll
      for (int k = 0; k < KS; ++k) dma_b32(bits_base + (long long)(s * KS + k) * (T::WAVES * kWave), bdst + 256 * k);
    };

    // MFMA lane roles (as in spmm_tc16_kernel): A row R of block g's 8 columns; B column R, rows 8g+q (+4)
    const int g = lane >> 4;
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int trow = 8 * g + q;
    const unsigned rd_off = trow * ROW_BYTES + 8 * p;
    const int tr_z = slot_swizzle<SLOTS>(trow);
    struct Frags {
      unsigned aw[KS];
      uint2_t blo[KS][SLOTS], bhi[KS][SLOTS];
    };
    auto read_frags = [&](int s, Frags& f) {      // LDS -> registers, asynchronous (lgkmcnt)
      const unsigned dt = data0 + (unsigned)(s % D) * STAGE_BYTES + rd_off;
      const unsigned bt = bits0 + (unsigned)(s % D) * T::BITS_BYTES + 4 * lane;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        f.aw[k] = lds_read_b32(bt + 256 * k);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
          const unsigned addr = dt + k * T::KSTEP_BYTES + ((sl ^ tr_z) << 5);
          if (VOLTRIX_PANEL_DIAG & 16) {
            f.blo[k][sl] = uint2_t{addr, addr};
            f.bhi[k][sl] = uint2_t{addr, addr};
            continue;
          }
          f.blo[k][sl] = lds_read_tr16_b64<0>(addr);
          f.bhi[k][sl] = lds_read_tr16_b64<4 * ROW_BYTES>(addr);
        }
      }
    };
    auto multiply = [&](int s, const Frags& f) {
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (s * KS + k < nks) {                   // workgroup-uniform (a panel's last group may be partial)
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const unsigned byte = (f.aw[k] >> (8 * j));
            const half8_t afrag = (VOLTRIX_PANEL_DIAG & 8)
                                      ? __builtin_bit_cast(half8_t, uint4_t{f.aw[k], f.aw[k], f.aw[k], f.aw[k]})
                                      : nibbles_to_half8_x2(byte & 0xFu, (byte >> 4) & 0xFu);
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
              const uint4_t bq = {f.blo[k][sl][0], f.blo[k][sl][1], f.bhi[k][sl][0], f.bhi[k][sl][1]};
              if (VOLTRIX_PANEL_DIAG & 1) {
                asm volatile("" ::"v"(afrag), "v"(bq));
                continue;
              }
              if constexpr (T::BF16)
                acc[j][sl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                     __builtin_bit_cast(bf16x8_t, bq), acc[j][sl], 0, 0, 0);
              else
                acc[j][sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq), acc[j][sl],
                                                                    0, 0, 0);
            }
          }
        }
      }
    };
    // counted wait: everything but the `young` most recently issued steps has landed
    auto wait_steps = [&](int young) {
      switch (young < D - 2 ? young : D - 2) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<T::VM_PER_STEP * 1>(); break;
        case 2: wait_vmcnt<T::VM_PER_STEP * 2>(); break;
        case 3: wait_vmcnt<T::VM_PER_STEP * 3>(); break;
        case 4: wait_vmcnt<T::VM_PER_STEP * 4>(); break;
        case 5: wait_vmcnt<T::VM_PER_STEP * 5>(); break;
        default: wait_vmcnt<T::VM_PER_STEP * 6>(); break;
      }
    };

    // ---- prologue: steps 0 .. D-2 in flight, fragments of step 0 in registers ------------------------------------
    const int npro = ngroups < D - 1 ? ngroups : D - 1;
    for (int s = 0; s < npro; ++s) issue_step(s, load_cols(s));
    Cols cnext = load_cols(D - 1 < ngroups ? D - 1 : 0);
    wait_steps(npro - 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags fa, fb;
    read_frags(0, fa);
    wait_lgkmcnt0();

    // One step: [rows of step t+1 landed] barrier; refill the slot step t-1 has left with step t+D-1; start reading
    // step t+1's fragments; multiply step t (read one step ago) while those reads and the DMAs are in flight.
    auto step = [&](int t, const Frags& cur, Frags& nxt) {
      const bool has_next = t + 1 < ngroups;      // workgroup-uniform
      if (has_next) {
        const int issued_last = t + D - 2 < ngroups - 1 ? t + D - 2 : ngroups - 1;
        wait_steps(issued_last - (t + 1));        // steps t+2 .. issued_last may stay in flight
        if (!(VOLTRIX_PANEL_DIAG & 4))
          __builtin_amdgcn_s_barrier();           // every wave's share of step t+1 landed; all reads of step t-1 done
        __builtin_amdgcn_sched_barrier(0);
        if (t + D - 1 < ngroups) {
          issue_step(t + D - 1, cnext);
          if (t + D < ngroups) cnext = load_cols(t + D);
        }
        read_frags(t + 1, nxt);
      }
      multiply(t, cur);
      wait_lgkmcnt0();                            // nxt (and cnext) have arrived
    };
    int t = 0;
    for (; t + 1 < ngroups; t += 2) {
      step(t, fa, fb);
      step(t + 1, fb, fa);
    }
    if (t < ngroups) step(t, fa, fb);
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

// Host launcher.  The plan arrays must be padded as the builder pads them: panel_cols by 2 k-steps (64 ints) and
// panel_bits by one k-step beyond S = panel_ptr[NP] (the metadata DMAs fetch 64 column ids at a time).
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

}  // namespace voltrix
