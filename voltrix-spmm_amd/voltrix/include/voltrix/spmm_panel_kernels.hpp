// Voltrix-SpMM for MI355X (gfx950) -- panel kernel: the "shared column" half of the two-level condensed format.
//
// The reference format (and spmm_tc16_kernel) condenses columns per 16-row window: a gathered row of B serves 16 rows
// of A and, on graphs with a few hundred edges per row, about ONE of them (TC-block fill 6-7 %): every edge costs one
// 2*F-byte row gather out of L2 / Infinity Cache, and that gather traffic -- not HBM, not the matrix cores -- is what
// bounds the kernel (profiles/HISTORY.md section 5).  Columns that are referenced by SEVERAL rows of a taller row panel (community /
// band structure, hub columns) can do better: gathered once per panel into LDS, shared by all the panel's windows.
//
//   panel       PANEL_ROWS = WAVES * RB * 16 consecutive rows (256 or 512); one workgroup per (panel, feature slab)
//   plan        per panel the sorted list of its shared columns (those with >= tau edges inside the panel; chosen by
//               the plan builder, panel_plan.hpp), cut into k-steps of 32 columns:
//                 panel_ptr  int32 [NP+1]            first k-step of every panel
//                 panel_cols int32 [32 * (S + pad)]  row of B per (k-step, k); unused slots repeat a real column
//                 panel_bits uint32 [(S + 1) * WAVES * 64] adjacency bits in MFMA A-operand order: word (k-step, wave v,
//                                                    lane L = 16 g + R), bit 16 (c & 1) + 4 j + (c >> 1)  <=>  edge (row 16 (RB v + j) + R
//                                                    of the panel, column 8 g + c of the k-step)
//   everything else (columns below tau) stays in the reference's window format and runs through spmm_tc16_kernel; this
//   kernel then adds its share onto C (accumulate = 1) -- two addends per element, so the sum does not depend on order.
//
// Per k-step the workgroup gathers 32 rows of B ONCE (8 KiB at FS = 128; LDS-DMA, every wave issues its share) and each
// wave multiplies it into RB 16-row blocks: RB * FS/16 v_mfma_f32_16x16x32_f16 per 2 * FS/16 transposed LDS reads.  The
// ring is shared, so there is one raw s_barrier per step: counted vmcnt wait -> barrier -> reads (cdna_hip_programming.md
// "Pipelining across barriers"); the metadata (every wave's 256 B of adjacency bits, the step's 32 rows) is fetched once per
// workgroup into a shared slot a ring ahead (PanelTile: three DMAs per k-step, issued by waves 0 .. 2).
//
// Bound: matrix cores (16 rows x 32 columns per MFMA at the panel's density), with 1/16 .. 1/32 of the window kernel's
// gather traffic per covered edge.  Being MFMA-bound is what makes it a good neighbour: on a second stream it overlaps the
// gather-bound window kernel on the same CUs (176 VGPRs x 2 waves per SIMD + 36 KB of LDS at DEPTH 3 leave exactly the 160
// registers and more than the 103 KB a (128, 3, 4) pair window workgroup needs: tests/test_register_budget.py).  The structured-sparse MFMAs were measured and are NOT used: they make this kernel
// faster alone and the pair slower (harness/experiments/smfmac_prototype/README.md).
//
// Inline-asm note: every LDS read here is asynchronous asm whose outputs are all consumed after the lgkmcnt wait.  A read
// with a dead output component (e.g. ds_read_b128 of three used words) lets the compiler hand that register out again
// while the read is still in flight; the late write-back then lands in its new owner.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "voltrix/spmm_kernels.hpp"

// Diagnostic builds only (-DVOLTRIX_EXPERIMENTAL, traits.hpp; harness/experiments/panel_diag.py, exp_corun_diag.py): bit 0 skips the MFMAs, bit 1 the row DMAs,
// bit 2 the barrier, bit 3 the fragment reads, bit 4 the metadata DMAs of the loop (bit 4 ONLY with bit 1: the row gathers take
// their row ids from the metadata slots).  Results are wrong by design; shipped kernels use 0.
#ifndef VOLTRIX_PANEL_DIAG
#define VOLTRIX_PANEL_DIAG 0
#endif
namespace voltrix {


//   FS     feature slab per workgroup (columns of B / C): 32, 64 or 128
//   DEPTH  ring slots (k-step groups of gathered rows in flight per workgroup)
//   WAVES  waves per workgroup (4 or 8)
//   RB     16-row blocks per wave (2 or 4): PANEL_ROWS = WAVES * RB * 16
//   KS     k-steps (of 32 columns) per ring slot / barrier
//   PIPE   the software-pipelined k-step loop (round 5, below): fragment reads of one half of the slots under the MFMAs of the
//          other half, the barrier and the refill between the halves
template <int FS_, int DEPTH_, int WAVES_, int RB_, int KS_ = 1, bool BF16_ = false, bool PIPE_ = false>
struct PanelTile {
  static constexpr int FS = FS_, DEPTH = DEPTH_, WAVES = WAVES_, RB = RB_, KS = KS_;
  static constexpr bool BF16 = BF16_, PIPE = PIPE_;
  static_assert(!PIPE || (KS == 1 && FS >= 32), "pipelined loop: one k-step per ring slot");
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
  // Metadata slot, shared by the workgroup (round 4; rounds 2-3: one slot per wave, every wave fetching its own 256 B of words
  // and its own copy of the SAME ids -- 16 small DMAs per k-step at 8 waves; now 3): the adjacency words of ALL waves for the
  // group's KS k-steps (word (k, wave v, lane L) at ((k WAVES + v) 64 + L) 4, as in panel_bits), then the 64 column ids.
  // The words come as 1-KiB dwordx4 DMAs issued by waves 0 .. NBITS_DMA - 1, the ids as one DMA of wave NBITS_DMA; the ring's
  // barrier per k-step makes them visible (a group's metadata lands D - 1 steps before its rows are issued).  Measured
  // (profiles/r04/experiment_meta_ab.log): headline step unchanged (the panel kernel gains what the window kernel loses),
  // block model -3.5 % (the panel kernel is its critical path), 8 KiB less LDS.
  static constexpr int BITS_BYTES = KS * WAVES * 256;
  static constexpr int META_BYTES = BITS_BYTES + 256;
  // classic loop: a group's metadata is issued 2 D - 2 steps ahead and dies with the group; pipelined loop: 2 D steps ahead
  // (its column ids are read one barrier earlier, its rows issued one step later) and the words die one step earlier
  static constexpr int META_SLOTS = PIPE ? 2 * DEPTH : 2 * DEPTH - 1;
  static constexpr int NBITS_DMA = BITS_BYTES / 1024;
  static_assert(BITS_BYTES % 1024 == 0 && NBITS_DMA + 1 <= WAVES, "metadata DMA roles");
  static constexpr int VM_PER_STEP = DPW + 1;                      // of waves 0 .. NBITS_DMA (one metadata DMA per step)
  static constexpr int VM_PER_STEP_PLAIN = DPW;                    // of the other waves
  static constexpr int DATA_LDS = DEPTH * STAGE_BYTES;
  static constexpr int BLOCK_LDS = DATA_LDS + META_SLOTS * META_BYTES;
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
  static_assert(VM_PER_STEP * (DEPTH - 2) <= 63, "vmcnt is a 6-bit counter on gfx9");
};

// Adjacency word -> MFMA A fragment.  Bit p (p = 4 j + r) of the word is column 2 r, bit 16 + p column 2 r + 1 of row
// block j, so one shift + one mask yields the packed fp16 pair {2.0 or 0.0} x 2 of register r (2.0 = 0x4000; the 0.5 is
// applied once in the epilogue, as in spmm_tc16_kernel).
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
  const int* panel_cols;       // [32 * (S + 2)]
  const uint32_t* panel_bits;  // [(S + 1) * WAVES * 64]
  const int* panel_order;      // optional: launch position -> panel (longest first); nullptr = natural
  const int* xcd_ptr;          // optional int32[9]: XCD x owns the launch positions [xcd_ptr[x], xcd_ptr[x + 1]) (ranges of equal
                               // work, hybrid.py::balance_xcd_ranges); nullptr: ranges of panels_per_xcd positions each
  const int* parts;            // optional int32 [num_panels = number of PARTS][4] = {panel, first k-step inside the panel, k-steps,
                               // slot}: launch position -> a bounded piece of a panel's k-step list (hybrid.py::panel_parts; replaces
                               // panel_order).  slot < 0: the panel is whole, its tile goes to C per `accumulate`; slot >= 0: the
                               // panel is cut, this piece's tile is STORED to partials[slot] and combine_panel_partials_kernel
                               // adds the pieces to C in slot order -- a fixed order whatever the pieces' timing
  float* partials;             // [slots][PANEL_ROWS][F] fp32 (per-call scratch; only with parts)
  const in_t* input;
  float* output;
  const float* out_scale;      // optional device scalar (see SpmmArgs::out_scale)
  int num_nodes;
  int num_panels;
  int panels_per_xcd;
  int F;
  int meta_nt;                 // 1: bitmap / column DMAs are non-temporal (launcher: one slab covers F, every byte read once)
  int slab_first;              // blockIdx.y counts column slabs from here (one launch per slab: launch_spmm_panel)
  int accumulate;              // 0: C = A_shared * B;  1: C += A_shared * B (C holds the window kernel's part, read-add-store);
                               // 2: C += A_shared * B by float atomics (C pre-zeroed, the window kernel adds its part the
                               //    same way, in any order: two addends per element, so the sum does not depend on it)
};

template <class T>
static __global__ __launch_bounds__(T::THREADS) void spmm_panel_kernel(const PanelArgs<T> a) {
  constexpr int FS = T::FS, D = T::DEPTH, MS = T::META_SLOTS, KS = T::KS, RB = T::RB;
  constexpr int ROW_BYTES = T::ROW_BYTES, STAGE_BYTES = T::STAGE_BYTES, DPW = T::DPW;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));

  // XCD x = blockIdx.x % 8 owns a contiguous range of launch positions: neighbouring panels share most of their
  // columns (band / community structure), so they should share an L2.
  const int xcd = blockIdx.x % kNumXcd;
  const int pos = (a.xcd_ptr ? a.xcd_ptr[xcd] : xcd * a.panels_per_xcd) + (int)(blockIdx.x / kNumXcd);
  const int pos_end = a.xcd_ptr ? a.xcd_ptr[xcd + 1]
                                : ((xcd + 1) * a.panels_per_xcd < a.num_panels ? (xcd + 1) * a.panels_per_xcd : a.num_panels);
  if (pos >= pos_end) return;  // workgroup-uniform
  int panel, ks0, nks, slot = -1;
  if (a.parts) {               // kernel-uniform
    const int4_t part = *reinterpret_cast<const int4_t*>(a.parts + 4 * (long long)pos);
    panel = part[0];
    ks0 = a.panel_ptr[panel] + part[1];
    nks = part[2];
    slot = part[3];
  } else {
    panel = a.panel_order ? a.panel_order[pos] : pos;
    ks0 = a.panel_ptr[panel];
    nks = a.panel_ptr[panel + 1] - ks0;
  }
  const int fs0 = (a.slab_first + blockIdx.y) * FS;
  const int F = a.F;
  const int lane = threadIdx.x & (kWave - 1);

  const int ngroups = (nks + KS - 1) / KS;
  if (ngroups == 0 && a.accumulate && slot < 0) return;  // workgroup-uniform: nothing to add

  float4_t acc[RB][SLOTS];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[j][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned data0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned meta0 = data0 + T::DATA_LDS;
  const bool meta_wave = wave <= T::NBITS_DMA;   // waves 0 .. NBITS_DMA - 1 fetch the words, wave NBITS_DMA the column ids

  if (ngroups > 0) {
    // ---- lane constants ---------------------------------------------------------------------------------------
    const unsigned row_bytes = (unsigned)F * 2u;
    const int dma0 = (wave * DPW) % T::NDMA;      // this wave's first row DMA of a step: LDS bytes [dma0 KiB, ...)
    const char* cbase[DPW];   // source of this lane's 16 bytes in row DMA d of a step, before the row offset
    unsigned hr_off[DPW];     // byte offset of that DMA's row id inside the metadata slot's column list
#pragma unroll
    for (int d = 0; d < DPW; ++d) {
      const int r = (dma0 + d) * RPD + lane / LPR;  // gathered row inside the step (0 .. 32 KS - 1)
      const int c = lane % LPR;                     // 16-byte chunk inside the row
      int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r & 31)) * 16) + (c & 1) * 8;  // swizzle on the SOURCE
      col = col < F ? col : fs0;                    // F % FS tail: stay in bounds, never stored
      unsigned long long cb = (unsigned long long)((const char*)a.input + (long long)col * 2);
      asm volatile("" : "+v"(cb));
      cbase[d] = (const char*)cb;
      hr_off[d] = T::BITS_BYTES + 4 * r;
    }
    // metadata DMAs of k-step group s (clamped to the panel's last group: the pipeline issues a static number of DMAs)
    // wave-uniform bases (scalar registers); the lane offset is added at each DMA from an opaque copy of the lane id, so
    // that no 64-bit per-lane pointer stays live across the k-step loop (4 VGPRs fewer: 179 of the 184 that fit beside a
    // window-kernel workgroup)
    const int* const cols_base = a.panel_cols + (long long)ks0 * kStageK;
    // ring positions are carried counters (ms = group % MS, ds = group % D), never a division: the k-step loop issued
    // 66 scalar instructions per step when they were computed with %
    auto issue_meta = [&](int s, int ms) {
      const int sc = s < ngroups ? s : ngroups - 1;
      const unsigned dst = meta0 + (unsigned)ms * T::META_BYTES;
      int ml = lane;
      asm volatile("" : "+v"(ml));
      if (wave < T::NBITS_DMA) {            // wave-uniform: 1 KiB of the group's KS x WAVES x 256 bytes of words
        const char* src = (const char*)(a.panel_bits + (long long)(ks0 + sc * KS) * (T::WAVES * kWave)) + wave * 1024 + ml * 16;
        if (a.meta_nt) dma_b128_nt(src, dst + wave * 1024); else dma_b128(src, dst + wave * 1024);
      } else if (wave == T::NBITS_DMA) {
        const int* src = cols_base + (long long)sc * (KS * kStageK) + ml;
        if (a.meta_nt) dma_b32_nt(src, dst + T::BITS_BYTES); else dma_b32(src, dst + T::BITS_BYTES);
      }
    };
    auto issue_rows = [&](int ms, int ds) {
      const unsigned mslot = meta0 + (unsigned)ms * T::META_BYTES;
      const unsigned dst = data0 + (unsigned)ds * STAGE_BYTES + (unsigned)dma0 * 1024u;
      unsigned hrow[DPW];
#pragma unroll
      for (int d = 0; d < DPW; ++d) hrow[d] = lds_read_b32(mslot + hr_off[d]);
      wait_lgkmcnt0();
#pragma unroll
      for (int d = 0; d < DPW; ++d)
        if (!(VOLTRIX_PANEL_DIAG & 2)) dma_b128(cbase[d] + (unsigned long long)hrow[d] * row_bytes, dst + d * 1024);
    };

    // ---- prologue: metadata of groups 0 .. D-2, then the virtual steps -(D-1) .. -1 ------------------------------
    // (pipelined loop: one group more of each -- its barrier comes AFTER a group's reads, so the ring holds D groups)
    constexpr int PRO = T::PIPE ? D : D - 1;
#pragma unroll
    for (int s = 0; s < PRO; ++s) issue_meta(s, s % MS);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();           // the other waves' metadata DMAs have landed too
#pragma unroll
    for (int s = 0; s < PRO; ++s) {
      issue_rows(s % MS, s % D);            // groups past the panel's end re-gather its last group (static DMA count)
      issue_meta(s + PRO, (s + PRO) % MS);
    }

    // MFMA lane roles (as in spmm_tc16_kernel): A row R of block g's 8 columns; B column R, rows 8g+q (+4).  Physical
    // slot of logical slot s = s ^ tr_z: address(s) = address(0) +- 32, +- 64, +- 128 per set bit of s, the sign being the
    // lane's -- SLOTS - 1 adds per k-step instead of an xor + add per slot
    const int g = lane >> 4;
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int trow = 8 * g + q;
    const int tr_z = slot_swizzle<SLOTS>(trow);
    unsigned rd_off = trow * ROW_BYTES + 8 * p + (tr_z << 5);
    int tr_delta[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      tr_delta[b] = ((tr_z >> b) & 1) ? -(32 << b) : (32 << b);
      asm volatile("" : "+v"(tr_delta[b]));
    }
    asm volatile("" : "+v"(rd_off));

    if constexpr (T::PIPE) {
      // ---- the software-pipelined k-step loop (round 5) -------------------------------------------------------------------
      // The classic loop below runs its phases one after the other in every wave -- barrier, LDS round trip for the row ids,
      // refill, LDS round trip for 16 fragment reads, 32 MFMAs -- and the eight waves do so in step, so the matrix cores idle
      // through everything that is not an MFMA (measured: the MFMA-free skeleton alone 0.22-0.33 ms of 0.81 on the headline
      // graph, MFMA busy 30 %).  Here a k-step's slots are two halves: the fragments of half 1 are read under the MFMAs of
      // half 0; then the barrier (every wave has read all of group t: its ring slot is refilled with group t + D) and the
      // fragment reads of half 0 of group t + 1 -- under the MFMAs of half 1.  The row id for the refill and the next group's
      // adjacency word ride along with fragment reads, so no LDS round trip stands alone.  Same accumulation order per
      // element (k-steps in order): bit-identical to the classic loop.
      constexpr int HS = SLOTS / 2 > 0 ? SLOTS / 2 : 1;     // slots per half
      constexpr int NH = SLOTS / HS;                          // halves (1 at FS = 16-column slabs: not instantiated)
      static_assert(NH == 2, "two halves");
      uint2_t b0lo[HS], b0hi[HS], b1lo[HS], b1hi[HS];
      // address of logical slot s: the ring is 256-byte aligned (checked below) and a row's slots are the address bits 5 .. 7,
      // so the physical slot s ^ z is one XOR with a constant -- no per-lane delta registers as in the classic loop
      if (data0 & 255u) __builtin_trap();   // wave-uniform; dynamic LDS starts at 0 in a kernel without static LDS
      auto half_addr = [&](const unsigned dt, const int s) -> unsigned { return dt ^ ((unsigned)s << 5); };
      auto read_half = [&](const int h, const unsigned dt, uint2_t (&lo)[HS], uint2_t (&hi)[HS]) {
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          const unsigned ad = half_addr(dt, h * HS + s);
          if (VOLTRIX_PANEL_DIAG & 8) {
            lo[s] = uint2_t{ad, 1u};
            hi[s] = uint2_t{ad, 2u};
            continue;
          }
          lo[s] = lds_read_tr16_b64<0>(ad);
          hi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(ad);
        }
      };
      auto mfma_half = [&](const int h, unsigned aw, const uint2_t (&lo)[HS], const uint2_t (&hi)[HS]) {
        // four row blocks: an opaque copy per half, so that the conversions are redone for the second half (8 VALU per row
        // block) instead of sixteen fragment registers staying live across the barrier (the 176-register budget of the pair)
        if constexpr (RB == 4) asm volatile("" : "+v"(aw));
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const half8_t afrag = adjacency_to_half8_x2(aw, 4 * j);
#pragma unroll
          for (int s = 0; s < HS; ++s) {
            const uint4_t bq = {lo[s][0], lo[s][1], hi[s][0], hi[s][1]};
            if (VOLTRIX_PANEL_DIAG & 1) {
              asm volatile("" ::"v"(afrag), "v"(bq));
              continue;
            }
            if constexpr (T::BF16)
              acc[j][h * HS + s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                             __builtin_bit_cast(bf16x8_t, bq), acc[j][h * HS + s], 0, 0, 0);
            else
              acc[j][h * HS + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq),
                                                                            acc[j][h * HS + s], 0, 0, 0);
          }
        }
      };
      auto issue_rows_of = [&](const unsigned (&hrow)[DPW], const int ds) {
        const unsigned dst = data0 + (unsigned)ds * STAGE_BYTES + (unsigned)dma0 * 1024u;
#pragma unroll
        for (int d = 0; d < DPW; ++d)
          if (!(VOLTRIX_PANEL_DIAG & 2)) dma_b128(cbase[d] + (unsigned long long)hrow[d] * row_bytes, dst + d * 1024);
      };
      // steps issued so far: the prologue's D, then one per iteration (after its barrier) while t + D < ngroups.  Before the
      // barrier of iteration t group t + 1 must have landed; younger: min(G - 2 - t, D - 2) steps, G = max(ngroups, D)
      const int total_steps = ngroups > D ? ngroups : D;
      auto wait_group = [&](const int young, auto per_step) {
        constexpr int VM = decltype(per_step)::value;
        if (young >= D - 2 && D >= 2) {
          wait_vmcnt<VM * (D - 2)>();
        } else {
          switch (young) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<VM * 1>(); break;
            case 2: wait_vmcnt<VM * 2>(); break;
            case 3: wait_vmcnt<VM * 3>(); break;
            case 4: wait_vmcnt<VM * 4>(); break;
            default: wait_vmcnt<VM * 5>(); break;
          }
        }
      };
      // group 0: the prologue issued D steps; D - 1 of them are younger than group 0
      if (meta_wave) wait_vmcnt<T::VM_PER_STEP * (D - 1)>();
      else wait_vmcnt<T::VM_PER_STEP_PLAIN * (D - 1)>();
      if (!(VOLTRIX_PANEL_DIAG & 4)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      unsigned aw_cur = lds_read_b32(meta0 + 256 * wave + 4 * lane);
      read_half(0, data0 + rd_off, b0lo, b0hi);
      wait_lgkmcnt0();
      int ds_t = 0;                      // t % D
      int ds_n = 1 % D;                  // (t + 1) % D
      int ms_t = 0;                      // t % MS == (t + 2 D) % MS: the slot the metadata issued this step goes to
      int ms_n = 1 % MS;                 // (t + 1) % MS
      int ms_d = D % MS;                 // (t + D) % MS
      for (int t = 0; t < ngroups; ++t) {
        const unsigned dt = data0 + (unsigned)ds_t * STAGE_BYTES + rd_off;
        const bool refill = t + D < ngroups;            // workgroup-uniform
        // half 1 of group t and the row ids of group t + D (landed D steps ago; visible since the last barrier)
        read_half(1, dt, b1lo, b1hi);
        unsigned hrow[DPW];
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
          // the id's address is rebuilt from the lane id here (two VALU operations) instead of living in a register through
          // the loop: the 8-wave tile sits exactly on the 176-register budget of the pair (tests/test_register_budget.py)
          unsigned ad;
          asm volatile("v_lshrrev_b32 %0, %2, %1\n\tv_lshl_add_u32 %0, %0, 2, %3"
                       : "=&v"(ad)
                       : "v"(lane), "n"(LPR == 16 ? 4 : (LPR == 8 ? 3 : 2)),
                         "s"(meta0 + (unsigned)ms_d * T::META_BYTES + T::BITS_BYTES + 4u * (unsigned)((dma0 + d) * RPD)));
          hrow[d] = lds_read_b32(ad);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(0, aw_cur, b0lo, b0hi);
        __builtin_amdgcn_sched_barrier(0);   // the MFMAs stay ABOVE the wait: they are what covers the reads' latency
        wait_lgkmcnt0();
#pragma unroll
        for (int d = 0; d < DPW; ++d) asm volatile("" ::"v"(hrow[d]));   // alive past the wait also when nothing is refilled
        unsigned aw_next = aw_cur;
        if (t + 1 < ngroups) {                          // workgroup-uniform
          const int young = total_steps - 2 - t;
          if (meta_wave) wait_group(young, std::integral_constant<int, T::VM_PER_STEP>{});
          else wait_group(young, std::integral_constant<int, T::VM_PER_STEP_PLAIN>{});
          if (!(VOLTRIX_PANEL_DIAG & 4))
            __builtin_amdgcn_s_barrier();   // group t + 1 has landed in every wave's share; everyone is done reading group t
          __builtin_amdgcn_sched_barrier(0);
          if (refill) {
            issue_rows_of(hrow, ds_t);      // group t + D into the slot group t has just left
            if (!(VOLTRIX_PANEL_DIAG & 16)) issue_meta(t + 2 * D, ms_t);
          }
          aw_next = lds_read_b32(meta0 + (unsigned)ms_n * T::META_BYTES + 256 * wave + 4 * lane);
          read_half(0, data0 + (unsigned)ds_n * STAGE_BYTES + rd_off, b0lo, b0hi);
          __builtin_amdgcn_sched_barrier(0);
        }
        mfma_half(1, aw_cur, b1lo, b1hi);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkmcnt0();
        aw_cur = aw_next;
        ds_t = ds_n;
        ds_n = ds_n + 1 == D ? 0 : ds_n + 1;
        ms_t = ms_n;
        ms_n = ms_n + 1 == MS ? 0 : ms_n + 1;
        ms_d = ms_d + 1 == MS ? 0 : ms_d + 1;
      }
    } else {
    int ds_t = 0, ms_t = 0;                              // group t
    int ds_r = (D - 1) % D, ms_r = (D - 1) % MS;         // group t + D - 1 (rows issued this step)
    int ms_m = (2 * D - 2) % MS;                         // group t + 2D - 2 (metadata issued this step)
    for (int t = 0; t < ngroups; ++t) {
      // rows of group t (issued D-1 steps ago) and the metadata of group t+D-1 must have landed; the D-2 younger
      // steps may stay in flight.  Steps past ngroups-D+1 issue nothing.
      const int young = ngroups - 1 - t;
      auto wait_steps = [&](auto per_step) {     // per_step: this wave's vector-memory operations per step (a constant)
        constexpr int VM = decltype(per_step)::value;
        if (young >= D - 2) {
          wait_vmcnt<VM*(D - 2)>();
        } else {
          switch (young) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<VM * 1>(); break;
            case 2: wait_vmcnt<VM * 2>(); break;
            case 3: wait_vmcnt<VM * 3>(); break;
            case 4: wait_vmcnt<VM * 4>(); break;
            default: wait_vmcnt<VM * 5>(); break;
          }
        }
      };
      if (meta_wave) wait_steps(std::integral_constant<int, T::VM_PER_STEP>{});
      else wait_steps(std::integral_constant<int, T::VM_PER_STEP_PLAIN>{});
      if (!(VOLTRIX_PANEL_DIAG & 4))
        __builtin_amdgcn_s_barrier();  // every wave's share of group t has landed; everyone is done reading group t-1
      __builtin_amdgcn_sched_barrier(0);

      if (t + D - 1 < ngroups) {     // workgroup-uniform
        issue_rows(ms_r, ds_r);      // into the slot group t-1 has just left
        if (!(VOLTRIX_PANEL_DIAG & 16)) issue_meta(t + 2 * D - 2, ms_m);   // diagnostic bit 16: no metadata DMAs in the loop
      }

      const unsigned mt = meta0 + (unsigned)ms_t * T::META_BYTES;
      const unsigned dt = data0 + (unsigned)ds_t * STAGE_BYTES + rd_off;
      ds_t = ds_t + 1 == D ? 0 : ds_t + 1;
      ds_r = ds_r + 1 == D ? 0 : ds_r + 1;
      ms_t = ms_t + 1 == MS ? 0 : ms_t + 1;
      ms_r = ms_r + 1 == MS ? 0 : ms_r + 1;
      ms_m = ms_m + 1 == MS ? 0 : ms_m + 1;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if (t * KS + k < nks) {      // workgroup-uniform
          const unsigned aw = lds_read_b32(mt + 256 * (k * T::WAVES + wave) + 4 * lane);
          unsigned taddr[SLOTS];
          taddr[0] = dt + k * T::KSTEP_BYTES;
#pragma unroll
          for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
            for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta[b];
          uint2_t blo[SLOTS], bhi[SLOTS];
#pragma unroll
          for (int s = 0; s < SLOTS; ++s) {
            if (VOLTRIX_PANEL_DIAG & 8) {   // diagnostic: no fragment reads
              blo[s] = uint2_t{taddr[s], 1u};
              bhi[s] = uint2_t{taddr[s], 2u};
              continue;
            }
            blo[s] = lds_read_tr16_b64<0>(taddr[s]);
            bhi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(taddr[s]);
          }
          wait_lgkmcnt0();
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const half8_t afrag = adjacency_to_half8_x2(aw, 4 * j);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
              const uint4_t bq = {blo[s][0], blo[s][1], bhi[s][0], bhi[s][1]};
              if (VOLTRIX_PANEL_DIAG & 1) {
                asm volatile("" ::"v"(afrag), "v"(bq));
                continue;
              }
              if constexpr (T::BF16)
                acc[j][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                    __builtin_bit_cast(bf16x8_t, bq), acc[j][s], 0, 0, 0);
              else
                acc[j][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq), acc[j][s], 0,
                                                                   0, 0);
            }
          }
        }
      }
    }
    }
    wait_vmcnt<0>();  // nothing of this workgroup may still be writing LDS when it is released
  }

  // ---- epilogue: D[row = 4*(lane>>4) + i][col = lane & 15] per (row block, 16-column slot) ------------------------
  const float oscale = kAScaleInv * (a.out_scale ? *a.out_scale : 1.0f);
  const int ocol0 = fs0 + (lane & 15);
  if (slot >= 0) {   // a piece of a cut panel: its tile goes to the partial slot, whole (rows past num_nodes hold zeros)
    float* const tile = a.partials + (long long)slot * T::PANEL_ROWS * F;
    const int trow0 = wave * (RB * 16) + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = ocol0 + 16 * s;
          if (col < F) tile[(long long)(trow0 + 16 * j + i) * F + col] = acc[j][s][i] * oscale;
        }
    return;
  }
  const int prow0 = panel * T::PANEL_ROWS + wave * (RB * 16) + 4 * (lane >> 4);
  if (a.accumulate == 2) {
#pragma unroll
    for (int j = 0; j < RB; ++j) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = prow0 + 16 * j + i, col = ocol0 + 16 * s;
          if (col < F && row < a.num_nodes) unsafeAtomicAdd(a.output + ((long long)row * F + col), acc[j][s][i] * oscale);
        }
      }
    }
    return;
  }
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

// Cut panels (PanelArgs::parts): C (+)= the pieces' tiles, summed in slot order.  One workgroup per (cut, 16 rows); a thread
// per float4 of the row, strided.  cuts int32 [C][4] = {panel, first slot, pieces, 0}.  HBM-bound, and small: only panels
// longer than the bound are cut.
static __global__ __launch_bounds__(256) void combine_panel_partials_kernel(const int* __restrict__ cuts,
                                                                            const float* __restrict__ partials,
                                                                            float* __restrict__ output, int num_nodes, int F,
                                                                            int panel_rows, int accumulate) {
  const int4_t cut = *reinterpret_cast<const int4_t*>(cuts + 4 * (long long)blockIdx.x);
  const int panel = cut[0], first = cut[1], pieces = cut[2];
  const int f4 = F / 4;
  const long long tile = (long long)panel_rows * F;
  for (int e = threadIdx.x; e < 16 * f4; e += 256) {
    const int r = blockIdx.y * 16 + e / f4, c = 4 * (e % f4);
    const long long row = (long long)panel * panel_rows + r;
    if (row >= num_nodes) continue;
    const float* src = partials + (long long)first * tile + (long long)r * F + c;
    float4_t sum = *reinterpret_cast<const float4_t*>(src);
    for (int j = 1; j < pieces; ++j) sum += *reinterpret_cast<const float4_t*>(src + j * tile);
    float4_t* dst = reinterpret_cast<float4_t*>(output + row * F + c);
    *dst = accumulate ? *dst + sum : sum;
  }
}

inline int launch_combine_panel_partials(const int* cuts, int num_cuts, const float* partials, float* output, int num_nodes,
                                         int embedding_dim, int panel_rows, int accumulate, hipStream_t stream) {
  if (num_cuts < 0 || num_nodes < 0 || embedding_dim < 0 || panel_rows < 16 || panel_rows % 16 != 0) return kErrBadShape;
  if (num_cuts == 0 || num_nodes == 0 || embedding_dim == 0) return kOk;
  if (embedding_dim % 4 != 0 || cuts == nullptr || partials == nullptr || output == nullptr) return kErrBadShape;
  hipLaunchKernelGGL(combine_panel_partials_kernel, dim3((unsigned)num_cuts, (unsigned)(panel_rows / 16)), dim3(256), 0, stream,
                     cuts, partials, output, num_nodes, embedding_dim, panel_rows, accumulate);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

// Host launcher.  The plan arrays must be padded as the builder pads them: panel_cols by 2 k-steps (64 ints) and
// panel_bits by one k-step beyond S = panel_ptr[NP] (the metadata DMAs fetch 64 column ids at a time).
template <class T>
inline int launch_spmm_panel(const int* panel_ptr, const int* panel_cols, const uint32_t* panel_bits,
                             const int* panel_order, int num_nodes, int embedding_dim, const void* input, float* output,
                             int accumulate, const float* out_scale, hipStream_t stream, int slab_first = 0,
                             int slab_count = 0 /* as launch_spmm_tc16: > 0 = that window of slabs in one launch */,
                             long long input_rows = 0 /* rows of the dense operand (0: num_nodes) */,
                             int slab_policy = kSlabAuto, const int* xcd_ptr = nullptr /* device int32[9] */,
                             int max_panels_per_xcd = 0 /* longest range of xcd_ptr (sizes the grid) */,
                             const int* parts = nullptr /* device int32[num_parts][4]: replaces panel_order */, int num_parts = 0,
                             float* partials = nullptr) {
  if (num_nodes < 0 || embedding_dim < 0 || accumulate < 0 || accumulate > 2) return kErrBadShape;
  if (parts != nullptr && num_parts < 1) return kErrBadShape;   // partials: required when any part carries a slot
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
  a.parts = parts;
  a.partials = partials;
  if (parts != nullptr) a.num_panels = num_parts;    // launch positions are parts
  a.panels_per_xcd = (a.num_panels + kNumXcd - 1) / kNumXcd;
  a.xcd_ptr = nullptr;
  if (xcd_ptr != nullptr) {
    if (max_panels_per_xcd < 1 || max_panels_per_xcd > a.num_panels) return kErrBadShape;
    a.xcd_ptr = xcd_ptr;
    a.panels_per_xcd = max_panels_per_xcd;      // sizes the grid; the kernel reads its range from xcd_ptr
  }
  a.F = embedding_dim;
  a.accumulate = accumulate;
  const int total_slabs = (embedding_dim + T::FS - 1) / T::FS;
  if (slab_first < 0 || slab_count < 0 || slab_first + slab_count > total_slabs) return kErrBadShape;
  if (const int group = slab_count == 0 ? slab_launch_group(total_slabs, T::ROW_BYTES,
                                                            input_rows > 0 ? input_rows : (long long)num_nodes, slab_policy)
                                        : 0) {   // spmm_kernels.hpp
    for (int s = 0; s < total_slabs; s += group) {
      const int rc = launch_spmm_panel<T>(panel_ptr, panel_cols, panel_bits, panel_order, num_nodes, embedding_dim, input,
                                          output, accumulate, out_scale, stream, s,
                                          total_slabs - s < group ? total_slabs - s : group, input_rows, slab_policy, xcd_ptr,
                                          max_panels_per_xcd, parts, num_parts, partials);
      if (rc != kOk) return rc;
    }
    return kOk;
  }
  const int slabs = slab_count > 0 ? slab_count : total_slabs;
  a.slab_first = slab_count > 0 ? slab_first : 0;
  a.meta_nt = total_slabs == 1;
  const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_panel_kernel<T>), T::BLOCK_LDS);
  if (lds_rc != kOk) return lds_rc;
  hipLaunchKernelGGL(spmm_panel_kernel<T>, dim3((unsigned)(a.panels_per_xcd * kNumXcd), (unsigned)slabs),
                     dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
