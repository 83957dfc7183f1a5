// Voltrix-SpMM for MI355X (gfx950) -- the two-level format in ONE launch (round 3; rebuilt in round 4).
//
// spmm_panel_kernel + spmm_tc16_pair_kernel run side by side on two streams and meet in C through a zero fill and two
// float-atomic epilogues (profiles/HISTORY.md section 3.3).  Here one workgroup owns a 512-row panel for the whole product: every wave
// keeps the accumulators of its 16-row blocks in registers and feeds them from BOTH halves of the format,
//
//   shared columns    the panel plan (panel_ptr / panel_cols / panel_bits, spmm_panel_kernels.hpp): 32 rows of B gathered
//                     once per k-step into a ring shared by the workgroup (one s_barrier per k-step), one
//                     v_mfma_f32_16x16x32_f16 per row block and 16-column slot;
//   residual columns  the edges the plan left in the reference's window format, re-packed per wave into a stream of
//                     256-byte STAGE RECORDS (32 condensed columns of ONE of the wave's windows: 32 rows of B, the 16 bitmap
//                     words, the row block), the wave's windows merged in column order so that its rows sweep the sorted
//                     columns together; consumed through a wave-private ring, no barrier;
//
// and C is written ONCE with plain stores: no zero fill, no atomics, no second stream, no combine pass, one fixed
// summation order per row (run-to-run identical bits).
//
// Round 4 geometry.  Round 3 ran 8 waves x 4 row blocks: two waves per SIMD, 128 accumulator registers
// each, and the residual as HALF stages (16 rows, 4 KiB, K = 16 MFMAs) through 8 rings of 3 x 4 KiB -- twice the
// wait -> read -> issue round trips per gathered byte of the window kernel, 30-37 GB/s per CU where that kernel does 55
// (2.03 ms against 1.35 ms for the pair).  Now: FOUR waves x EIGHT row blocks -- one wave per SIMD with the whole register
// file (256 accumulator registers in the AGPR half, the rest for fragments and addresses), the residual as WHOLE stages
// (32 rows, 8 KiB, K = 32 MFMAs: the window kernel's own inner loop) through 4 rings of 3 x 8 KiB, every B fragment of a
// panel k-step read from LDS once per 8 row blocks instead of once per 4.  The plan keeps its 8 x 4 layout: wave v reads the
// adjacency words of plan waves 2 v and 2 v + 1.
//
// Accumulators and the compiler.  A record's row block is data, the accumulators are registers: the residual MFMAs pick
// one of eight accumulator sets at run time.  Every MFMA of this kernel is inline asm with the accumulator tied in place
// ("+a": the AGPR file), the eight-way scalar branch of the residual sits INSIDE one asm statement that ties all eight
// candidate sets -- no control flow the register allocator can see, no copies (a C++ switch around the builtin made hipcc
// copy every accumulator per step; mixing builtin MFMAs with "+a" asm made it shuttle them between the two halves of the
// register file).  Hazards the compiler cannot see are padded inside the strings (s_nop 1 after the VALU-written A operand).
//
// vmcnt bookkeeping.  The two pipelines issue a data-dependent mix of LDS-DMAs, so no wait count is a compile-time
// constant.  Every wave counts the vector-memory operations it has issued (`nops`, a scalar) and remembers the count
// after each group it will wait for (`mark`); "that group has landed" is then s_waitcnt vmcnt(nops - mark), EXACT, through
// the jump table of wait_vm (spmm_kernels.hpp).  Nothing else in the loop may issue vector memory operations.
//
// LDS (FS = 128, DP = 3): panel ring 24 KiB + panel metadata 4 x 5 x 768 B + residual rings 4 x 24 KiB + records 4 x 2 KiB
// = 143 KiB: one workgroup per CU.
#pragma once

#include <mutex>

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_panel_kernels.hpp"

// Diagnostic builds only (-DVOLTRIX_EXPERIMENTAL, traits.hpp; harness/experiments/exp_fused_diag.py): bit 0 drops the residual
// steps, bit 1 the panel loop (the whole residual then runs barrier-free), bit 2 folds every residual row into the first 1024
// rows of B (all L2 hits).  Results are wrong by design; shipped kernels use 0.
#ifndef VOLTRIX_FUSED_DIAG
#define VOLTRIX_FUSED_DIAG 0
#endif

namespace voltrix {

constexpr int kFusedWaves = 4;          // waves per workgroup: one per SIMD
constexpr int kFusedRowBlocks = 8;      // 16-row blocks (= windows) per wave
constexpr int kFusedPanelRows = kFusedWaves * kFusedRowBlocks * kBlkH;   // 512 = the plan's 8 x 4 x 16
constexpr int kFusedPlanWaves = 8;      // the panel plan's own geometry (panel_bits: one word per plan wave and lane)
constexpr int kRecordWords = 64;        // a residual stage record: 32 hind | 16 bitmap words | row block | pad
constexpr int kRecordBytes = 4 * kRecordWords;
constexpr int kRecordBlockWord = 48;    // word holding the record's row block (0 .. 7) inside its wave

//   FS     feature slab per workgroup (32, 64 or 128 columns of B / C)
//   DP     slots of the shared panel ring (k-steps of gathered rows in flight per workgroup): 3 or 4
template <int FS_, int DP_ = 3, bool BF16_ = false>
struct FusedTile {
  static constexpr int FS = FS_, DP = DP_;
  static constexpr bool BF16 = BF16_;
  static_assert(FS == 32 || FS == 64 || FS == 128, "feature slab");
  static_assert(DP >= 3 && DP <= 4, "panel ring depth");
  static constexpr int WAVES = kFusedWaves, RB = kFusedRowBlocks;
  static constexpr int THREADS = WAVES * kWave;
  static constexpr int ROW_BYTES = FS * 2;
  static constexpr int SLOTS = FS / 16;
  static constexpr int LANES_PER_ROW = ROW_BYTES / 16;
  static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;
  // ---- panel half ----
  static constexpr int KSTEP_BYTES = kStageK * ROW_BYTES;           // 32 gathered rows
  static constexpr int NDMA_P = KSTEP_BYTES / 1024;                 // row DMAs per k-step, shared out over the waves
  static_assert(NDMA_P % WAVES == 0 || WAVES % NDMA_P == 0, "row DMAs per k-step vs waves");
  static constexpr int DPW = NDMA_P >= WAVES ? NDMA_P / WAVES : 1;  // per wave (surplus waves repeat the first ones)
  static constexpr int META_P_BYTES = 768;                          // 2 x 64 adjacency words + 64 column ids (32 used)
  static constexpr int MSP = 2 * DP - 1;                            // metadata slots per wave
  static constexpr int VM_PER_KSTEP = DPW + 3;
  static constexpr int DATA_P = DP * KSTEP_BYTES;
  // ---- residual half: whole stages of 32 gathered rows ----
  static constexpr int DR = 3;                                      // stage slots in the wave-private ring
  static constexpr int STAGE_BYTES = KSTEP_BYTES;
  static constexpr int NDMA_R = STAGE_BYTES / 1024;
  static constexpr int REC_AHEAD = DR;                              // record of stage h + DR + REC_AHEAD is fetched at step h
  static constexpr int MSR = 2 * DR + 2;                            // record slots per wave
  static constexpr int OFF_META_P = DATA_P;
  static constexpr int OFF_RING_R = OFF_META_P + WAVES * MSP * META_P_BYTES;
  static constexpr int OFF_META_R = OFF_RING_R + WAVES * DR * STAGE_BYTES;
  static constexpr int BLOCK_LDS = OFF_META_R + WAVES * MSR * kRecordBytes;
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
  // most operations a wave can have in flight: DP - 1 k-steps of the panel pipeline + DR stages and 2 DR records
  static_assert((DP - 1) * VM_PER_KSTEP + DR * NDMA_R + 2 * DR + 2 <= 63, "vmcnt is a 6-bit counter on gfx9");
};

// ---- MFMAs as inline asm on the accumulator file ------------------------------------------------------------------------
// acc[s] += A x B[s] for the NS 16-column slots of ONE row block.  s_nop 1: the A operand may have been written by the VALU
// instruction just before the statement (cdna_hip_programming.md section 5.7 item 2).
#define VOLTRIX_MF(OP, I) OP " %" #I ", %[a], %[b" #I "], %" #I "\n"
template <bool BF16>
__device__ __forceinline__ void mfma_block(float4_t (&acc)[8], const half8_t a, const uint4_t (&b)[8]) {
#define VOLTRIX_BODY(OP)                                                                                                  \
  asm volatile("s_nop 1\n" VOLTRIX_MF(OP, 0) VOLTRIX_MF(OP, 1) VOLTRIX_MF(OP, 2) VOLTRIX_MF(OP, 3) VOLTRIX_MF(OP, 4)      \
               VOLTRIX_MF(OP, 5) VOLTRIX_MF(OP, 6) VOLTRIX_MF(OP, 7)                                                       \
               : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]) \
               : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), \
                 [b6] "v"(b[6]), [b7] "v"(b[7]))
  if constexpr (BF16)
    VOLTRIX_BODY("v_mfma_f32_16x16x32_bf16");
  else
    VOLTRIX_BODY("v_mfma_f32_16x16x32_f16");
#undef VOLTRIX_BODY
}
template <bool BF16>
__device__ __forceinline__ void mfma_block(float4_t (&acc)[4], const half8_t a, const uint4_t (&b)[4]) {
#define VOLTRIX_BODY(OP)                                                                                                  \
  asm volatile("s_nop 1\n" VOLTRIX_MF(OP, 0) VOLTRIX_MF(OP, 1) VOLTRIX_MF(OP, 2) VOLTRIX_MF(OP, 3)                        \
               : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3])                                                   \
               : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]))
  if constexpr (BF16)
    VOLTRIX_BODY("v_mfma_f32_16x16x32_bf16");
  else
    VOLTRIX_BODY("v_mfma_f32_16x16x32_f16");
#undef VOLTRIX_BODY
}
template <bool BF16>
__device__ __forceinline__ void mfma_block(float4_t (&acc)[2], const half8_t a, const uint4_t (&b)[2]) {
#define VOLTRIX_BODY(OP)                                                                                                  \
  asm volatile("s_nop 1\n" VOLTRIX_MF(OP, 0) VOLTRIX_MF(OP, 1)                                                            \
               : "+a"(acc[0]), "+a"(acc[1])                                                                               \
               : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]))
  if constexpr (BF16)
    VOLTRIX_BODY("v_mfma_f32_16x16x32_bf16");
  else
    VOLTRIX_BODY("v_mfma_f32_16x16x32_f16");
#undef VOLTRIX_BODY
}
#undef VOLTRIX_MF

// acc[j][i] += A x B[i], i < 4, for a wave-uniform run-time row block j in 0 .. 7: the eight-way scalar branch inside one
// statement that ties all 32 candidate accumulators (operand number 4 j + i = c[j][i]).  The MFMAs of a statement touch
// different accumulators; the next reader of any of them is at least one s_waitcnt away.
template <bool BF16>
__device__ __forceinline__ void mfma_select8(float4_t (&c)[8][4], const half8_t a, const uint4_t (&b)[4], const int j) {
#define VOLTRIX_MS(OP, C, B) OP " %" #C ", %[a], %[" #B "], %" #C "\n"
#define VOLTRIX_ROW(OP, C0, C1, C2, C3) VOLTRIX_MS(OP, C0, b0) VOLTRIX_MS(OP, C1, b1) VOLTRIX_MS(OP, C2, b2) VOLTRIX_MS(OP, C3, b3)
#define VOLTRIX_BODY(OP)                                                                                                  \
  asm volatile("s_nop 1\n"                                                                                                \
               "s_cmp_lt_u32 %[j], 4\n s_cbranch_scc1 14f\n"                                                              \
               "s_cmp_lt_u32 %[j], 6\n s_cbranch_scc1 16f\n"                                                              \
               "s_cmp_eq_u32 %[j], 6\n s_cbranch_scc1 26f\n"                                                              \
               VOLTRIX_ROW(OP, 28, 29, 30, 31) "s_branch 99f\n"                                                           \
               "26:\n" VOLTRIX_ROW(OP, 24, 25, 26, 27) "s_branch 99f\n"                                                   \
               "16:\n s_cmp_eq_u32 %[j], 4\n s_cbranch_scc1 24f\n"                                                        \
               VOLTRIX_ROW(OP, 20, 21, 22, 23) "s_branch 99f\n"                                                           \
               "24:\n" VOLTRIX_ROW(OP, 16, 17, 18, 19) "s_branch 99f\n"                                                   \
               "14:\n s_cmp_lt_u32 %[j], 2\n s_cbranch_scc1 12f\n"                                                        \
               "s_cmp_eq_u32 %[j], 2\n s_cbranch_scc1 22f\n"                                                              \
               VOLTRIX_ROW(OP, 12, 13, 14, 15) "s_branch 99f\n"                                                           \
               "22:\n" VOLTRIX_ROW(OP, 8, 9, 10, 11) "s_branch 99f\n"                                                     \
               "12:\n s_cmp_eq_u32 %[j], 0\n s_cbranch_scc1 20f\n"                                                        \
               VOLTRIX_ROW(OP, 4, 5, 6, 7) "s_branch 99f\n"                                                               \
               "20:\n" VOLTRIX_ROW(OP, 0, 1, 2, 3)                                                                        \
               "99:\n"                                                                                                    \
               : "+a"(c[0][0]), "+a"(c[0][1]), "+a"(c[0][2]), "+a"(c[0][3]), "+a"(c[1][0]), "+a"(c[1][1]), "+a"(c[1][2]), \
                 "+a"(c[1][3]), "+a"(c[2][0]), "+a"(c[2][1]), "+a"(c[2][2]), "+a"(c[2][3]), "+a"(c[3][0]), "+a"(c[3][1]), \
                 "+a"(c[3][2]), "+a"(c[3][3]), "+a"(c[4][0]), "+a"(c[4][1]), "+a"(c[4][2]), "+a"(c[4][3]), "+a"(c[5][0]), \
                 "+a"(c[5][1]), "+a"(c[5][2]), "+a"(c[5][3]), "+a"(c[6][0]), "+a"(c[6][1]), "+a"(c[6][2]), "+a"(c[6][3]), \
                 "+a"(c[7][0]), "+a"(c[7][1]), "+a"(c[7][2]), "+a"(c[7][3])                                               \
               : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [j] "s"(j)                    \
               : "scc")
  if constexpr (BF16)
    VOLTRIX_BODY("v_mfma_f32_16x16x32_bf16");
  else
    VOLTRIX_BODY("v_mfma_f32_16x16x32_f16");
#undef VOLTRIX_BODY
#undef VOLTRIX_ROW
#undef VOLTRIX_MS
}

template <class T>
struct FusedArgs {
  using in_t = typename std::conditional<T::BF16, bfloat16_bits, _Float16>::type;
  const int* panel_ptr;        // [NP+1]
  const int* panel_cols;       // [32 * (S + 2)]
  const uint32_t* panel_bits;  // [(S + 1) * 8 * 64]   (the plan's 8 x 4 layout)
  const int* panel_order;      // optional: launch position -> panel (longest first); nullptr = natural
  const int* xcd_ptr;          // optional int32[9]: XCD x owns the launch positions [xcd_ptr[x], xcd_ptr[x + 1]) (PanelArgs::xcd_ptr)
  const int* wave_ptr;         // [4 NP + 1]: first residual stage record of (panel, wave)
  const uint32_t* records;     // [R + 1][64]
  const in_t* input;
  float* output;
  const float* out_scale;      // optional device scalar (SpmmArgs::out_scale)
  int num_nodes;
  int num_panels;
  int panels_per_xcd;
  int F;
  int meta_nt;                 // 1: metadata DMAs are non-temporal (one slab covers F: every byte is read once)
  int* pace;                   // optional: zeroed int32 [8][kPaceGens][kPaceBlocks] arrival counters (launcher).  The workgroups
                               // that share an XCD label and a dispatch generation wait for each other -- bounded, advisory:
                               // correctness never depends on it -- at pace_blocks points of their column sweep, so that an
                               // XCD's 16 k resident rows sweep the columns together (the L2-hit lever, profiles/HISTORY.md section 3.7)
  int pace_blocks;
};
constexpr int kPaceGens = 8, kPaceBlocks = 64;

template <class T>
static __global__ __launch_bounds__(T::THREADS) void spmm_fused_kernel(const FusedArgs<T> a) {
  constexpr int DP = T::DP, MSP = T::MSP, DR = T::DR, MSR = T::MSR, RB = T::RB, FS = T::FS;
  constexpr int ROW_BYTES = T::ROW_BYTES, KSTEP_BYTES = T::KSTEP_BYTES, DPW = T::DPW, STAGE_BYTES = T::STAGE_BYTES;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS, NDMA_R = T::NDMA_R;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int lane = threadIdx.x & (kWave - 1);

  // XCD x = blockIdx.x % 8 owns a contiguous range of launch positions (as spmm_panel_kernel)
  const int xcd = blockIdx.x % kNumXcd;
  const int pos0 = a.xcd_ptr ? a.xcd_ptr[xcd] : xcd * a.panels_per_xcd;
  const int pos = pos0 + (int)(blockIdx.x / kNumXcd);
  const int pos_end = a.xcd_ptr ? a.xcd_ptr[xcd + 1]
                                : ((xcd + 1) * a.panels_per_xcd < a.num_panels ? (xcd + 1) * a.panels_per_xcd : a.num_panels);
  if (pos >= pos_end) return;  // workgroup-uniform
  const int panel = a.panel_order ? a.panel_order[pos] : pos;
  const int fs0 = blockIdx.y * FS;
  const int F = a.F;

  const int ks0 = a.panel_ptr[panel];
  const int nks = (VOLTRIX_FUSED_DIAG & 2) ? 0 : a.panel_ptr[panel + 1] - ks0;
  const int rec0 = a.wave_ptr[panel * T::WAVES + wave];
  const int H = (VOLTRIX_FUSED_DIAG & 1) ? 0 : a.wave_ptr[panel * T::WAVES + wave + 1] - rec0;   // residual stages of this wave

  float4_t acc[RB][SLOTS];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[j][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned data_p = lds0;
  const unsigned meta_p = lds0 + T::OFF_META_P + (unsigned)wave * (MSP * T::META_P_BYTES);
  const unsigned ring_r = lds0 + T::OFF_RING_R + (unsigned)wave * (DR * STAGE_BYTES);
  const unsigned meta_r = lds0 + T::OFF_META_R + (unsigned)wave * (MSR * kRecordBytes);

  const unsigned row_bytes = (unsigned)F * 2u;
  const int g = lane >> 4, R = lane & 15;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3;

  int nops = 0;   // vector-memory operations issued by this wave so far (wave-uniform)

  // ======================================================= panel half: lane constants, issue helpers ==============
  const int dma0 = (wave * DPW) % T::NDMA_P;
  const char* cbase_p[DPW];
  unsigned hr_off_p[DPW];
#pragma unroll
  for (int d = 0; d < DPW; ++d) {
    const int r = (dma0 + d) * RPD + lane / LPR;
    const int c = lane % LPR;
    int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r & 31)) * 16) + (c & 1) * 8;  // swizzle on the SOURCE
    col = col < F ? col : fs0;
    unsigned long long cb = (unsigned long long)((const char*)a.input + (long long)col * 2);
    asm volatile("" : "+v"(cb));
    cbase_p[d] = (const char*)cb;
    hr_off_p[d] = 512 + 4 * r;
  }
  // adjacency words of plan waves 2 wave, 2 wave + 1 (the plan is laid out for 8 waves x 4 row blocks)
  const uint32_t* const bits_base = a.panel_bits + ((long long)ks0 * kFusedPlanWaves + 2 * wave) * kWave;
  const int* const cols_base = a.panel_cols + (long long)ks0 * kStageK;
  auto issue_meta_p = [&](const int s, const int ms) {   // k-step s < nks: 3 operations
    const unsigned dst = meta_p + (unsigned)ms * T::META_P_BYTES;
    int ml = lane;
    asm volatile("" : "+v"(ml));
    const uint32_t* const bsrc = bits_base + (long long)s * (kFusedPlanWaves * kWave) + ml;
    if (a.meta_nt) {
      dma_b32_nt(bsrc, dst);
      dma_b32_nt(bsrc + kWave, dst + 256);
      dma_b32_nt(cols_base + (long long)s * kStageK + ml, dst + 512);
    } else {
      dma_b32(bsrc, dst);
      dma_b32(bsrc + kWave, dst + 256);
      dma_b32(cols_base + (long long)s * kStageK + ml, dst + 512);
    }
    nops += 3;
  };
  auto issue_rows_p = [&](const int ms, const int ds) {   // DPW operations
    const unsigned mslot = meta_p + (unsigned)ms * T::META_P_BYTES;
    const unsigned dst = data_p + (unsigned)ds * KSTEP_BYTES + (unsigned)dma0 * 1024u;
    unsigned hrow[DPW];
#pragma unroll
    for (int d = 0; d < DPW; ++d) hrow[d] = lds_read_b32(mslot + hr_off_p[d]);
    wait_lgkmcnt0();
#pragma unroll
    for (int d = 0; d < DPW; ++d) dma_b128(cbase_p[d] + (unsigned long long)hrow[d] * row_bytes, dst + d * 1024);
    nops += DPW;
  };
  // transposed B reads (both rings hold the same 32-row image: spmm_kernels.hpp "LDS image")
  const int trow = 8 * g + q4;
  const int tr_z = slot_swizzle<SLOTS>(trow);
  unsigned rd_off = trow * ROW_BYTES + 8 * p4 + (tr_z << 5);
  int tr_delta[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tr_delta[b] = ((tr_z >> b) & 1) ? -(32 << b) : (32 << b);
    asm volatile("" : "+v"(tr_delta[b]));
  }
  asm volatile("" : "+v"(rd_off));
  // 2 SLOTS asynchronous LDS reads; the fragments are assembled by the caller AFTER its lgkmcnt wait
  auto read_b_fragments = [&](const unsigned slot_base, uint2_t (&blo)[SLOTS], uint2_t (&bhi)[SLOTS]) {
    unsigned taddr[SLOTS];
    taddr[0] = slot_base + rd_off;
#pragma unroll
    for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
      for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta[b];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      blo[s] = lds_read_tr16_b64<0>(taddr[s]);
      bhi[s] = lds_read_tr16_b64<4 * ROW_BYTES>(taddr[s]);
    }
  };

  // ======================================================= residual half: lane constants, issue helpers ===========
  // row DMA i of a stage writes LDS rows i * RPD + lane / LPR; the lane's 16 bytes come from column chunk (lane % LPR) of that
  // row of B, slot-swizzled on the source.  DMAs i and i + 4 (FS = 128) differ by 16 rows: same swizzle, same base pointer.
  constexpr int NBASE = (NDMA_R == 8 && RPD == 4) ? 4 : NDMA_R;
  static_assert(NBASE == NDMA_R || slot_swizzle<SLOTS>(4 * RPD + 1) == slot_swizzle<SLOTS>(1), "period of the row swizzle");
  const char* cbase_r[NBASE];
#pragma unroll
  for (int i = 0; i < NBASE; ++i) {
    const int r = i * RPD + lane / LPR;
    const int c = lane % LPR;
    int col = fs0 + (((c >> 1) ^ slot_swizzle<SLOTS>(r)) * 16) + (c & 1) * 8;
    col = col < F ? col : fs0;
    unsigned long long cb = (unsigned long long)((const char*)a.input + ((long long)col * 2 - (i & 3) * 1024));
    asm volatile("" : "+v"(cb));
    cbase_r[i] = (const char*)cb;
  }
  const unsigned id_off_r = 4 * (lane / LPR);   // + 4 RPD i: the lane's row of DMA i inside the record
  // A fragment: lane (R, g) holds row R of TC block g: nibble R & 7 of bitmap words 4 g + (R >> 3) and 4 g + 2 + (R >> 3)
  const unsigned aw_off_r = 128 + 4 * (4 * g + (R >> 3));
  const unsigned a_shift = 4 * (R & 7);

  const uint32_t* const rec_base = a.records + (long long)rec0 * kRecordWords;
  auto issue_record = [&](const int s, const int ms) {    // record s < H: 1 operation
    int ml = lane;
    asm volatile("" : "+v"(ml));
    const uint32_t* const src = rec_base + (long long)s * kRecordWords + ml;
    if (a.meta_nt)
      dma_b32_nt(src, meta_r + (unsigned)ms * kRecordBytes);
    else
      dma_b32(src, meta_r + (unsigned)ms * kRecordBytes);
    nops += 1;
  };
  auto read_ids = [&](const int ms, int (&ids)[NDMA_R]) {   // the lane's rows of B for the NDMA_R DMAs of the record in slot ms
    lds_read_b32_strided<4 * RPD>(meta_r + (unsigned)ms * kRecordBytes + id_off_r, ids, std::make_integer_sequence<int, NDMA_R>{});
  };
  auto issue_stage = [&](const int rs, const int (&ids)[NDMA_R]) {   // NDMA_R operations
    const unsigned dst = ring_r + (unsigned)rs * STAGE_BYTES;
    auto piece = [&](auto kc, int ib) {           // DMA number ib + K of the stage, K = 0 .. 3 sharing one M0
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDMA_R) {
        const int i = ib + K;
        unsigned hrow = (unsigned)ids[i];
        if (VOLTRIX_FUSED_DIAG & 4) hrow &= 1023u;
        const char* src = cbase_r[i % NBASE] + (unsigned long long)hrow * row_bytes;
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (lds_ptr)(uintptr_t)(dst + ib * 1024), 16, K * 1024, 0);
      }
    };
#pragma unroll
    for (int ib = 0; ib < NDMA_R; ib += 4) {
      piece(std::integral_constant<int, 0>{}, ib);
      piece(std::integral_constant<int, 1>{}, ib);
      piece(std::integral_constant<int, 2>{}, ib);
      piece(std::integral_constant<int, 3>{}, ib);
    }
    nops += NDMA_R;
  };

  // ======================================================= prologue ===============================================
  // panel: metadata of k-steps 0 .. DP-2; residual: records 0 .. 2 DR - 1; then the first rows of both rings
#pragma unroll
  for (int s = 0; s < DP - 1; ++s)
    if (s < nks) issue_meta_p(s, s % MSP);
#pragma unroll
  for (int s = 0; s < 2 * DR; ++s)
    if (s < H) issue_record(s, s % MSR);
  wait_vmcnt<0>();
  __builtin_amdgcn_sched_barrier(0);

  int mark_p[DP - 1];   // mark_p[i]: nops after the panel issues of the iteration that is DP-1-i iterations back
#pragma unroll
  for (int s = 0; s < DP - 1; ++s) {
    if (s < nks) issue_rows_p(s % MSP, s % DP);
    if (s + DP - 1 < nks) issue_meta_p(s + DP - 1, (s + DP - 1) % MSP);
    mark_p[s] = nops;
  }
  int mark_r[DR];       // mark_r[i]: nops after the refill that filled the stage consumed i steps from now
#pragma unroll
  for (int h0 = 0; h0 < DR; ++h0) {
    if (h0 < H) {
      int ids[NDMA_R];
      read_ids(h0 % MSR, ids);
      wait_lgkmcnt0();
      issue_stage(h0, ids);
    }
    mark_r[h0] = nops;
  }

  // ======================================================= one residual step = one whole stage ====================
  // Invariant at the top of step h: the rows of stages h .. h + DR - 1 and the records up to h + 2 DR - 1 are issued (as far as
  // they exist), and every record r was issued BEFORE the rows of stage r - DR -- so waiting for the rows of stage h (the
  // oldest thing this step needs) also covers the record of stage h + DR, whose rows this step issues.  Step h therefore
  // issues, in this order, record h + 2 DR (into the slot record h - 2 left two steps ago) and then the rows of stage h + DR
  // (into the ring slot it has just read).
  int h = 0;            // next stage
  int rs_h = 0;         // h % DR
  int ms_h = 0;         // h % MSR
  auto resid_step = [&]() {
    wait_vm(nops - mark_r[0]);
    const unsigned rec = meta_r + (unsigned)ms_h * kRecordBytes;
    const unsigned jw = lds_read_b32(rec + 4 * kRecordBlockWord);
    const unsigned wlo = lds_read_b32(rec + aw_off_r);
    const unsigned whi = lds_read_b32(rec + aw_off_r + 8);
    const bool refill = h + DR < H;            // wave-uniform
    int ids[NDMA_R];
    read_ids((ms_h + DR) % MSR, ids);          // unconditional (a conditional read makes the compiler zero-fill ids[] every step;
                                               // past the end the slot holds an older record: finite ids, never used)
    uint2_t blo[SLOTS], bhi[SLOTS];
    read_b_fragments(ring_r + (unsigned)rs_h * STAGE_BYTES, blo, bhi);
    wait_lgkmcnt0();
    if (h + 2 * DR < H) issue_record(h + 2 * DR, (ms_h + 2 * DR) % MSR);
    if (refill) issue_stage(rs_h, ids);
#pragma unroll
    for (int i = 0; i < DR - 1; ++i) mark_r[i] = mark_r[i + 1];
    mark_r[DR - 1] = nops;

    const half8_t afrag = nibbles_to_half8_x2((wlo >> a_shift) & 0xFu, (whi >> a_shift) & 0xFu);
    const int j = __builtin_amdgcn_readfirstlane((int)jw) & 7;
    uint4_t bq[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) bq[s] = uint4_t{blo[s][0], blo[s][1], bhi[s][0], bhi[s][1]};
    if constexpr (SLOTS >= 4) {
#pragma unroll
      for (int s0 = 0; s0 < SLOTS; s0 += 4) {
        float4_t c[8][4];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) c[jj][i] = acc[jj][s0 + i];
        const uint4_t b4[4] = {bq[s0], bq[s0 + 1], bq[s0 + 2], bq[s0 + 3]};
        mfma_select8<T::BF16>(c, afrag, b4, j);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[jj][s0 + i] = c[jj][i];
      }
    } else {   // FS = 32: two slots; the second pair of every row is a scratch copy
      float4_t c[8][4];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        c[jj][0] = acc[jj][0];
        c[jj][1] = acc[jj][1];
        c[jj][2] = c[jj][3] = float4_t{0.f, 0.f, 0.f, 0.f};
      }
      const uint4_t b4[4] = {bq[0], bq[1], bq[0], bq[1]};
      mfma_select8<T::BF16>(c, afrag, b4, j);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        acc[jj][0] = c[jj][0];
        acc[jj][1] = c[jj][1];
      }
    }
    ++h;
    rs_h = rs_h + 1 == DR ? 0 : rs_h + 1;
    ms_h = ms_h + 1 == MSR ? 0 : ms_h + 1;
  };

  // ======================================================= panel loop: one k-step + this wave's share of stages ====
  if (nks > 0) {
    // stages spread evenly over the k-steps: q_t = base (+ 1 whenever the remainder accumulator wraps)
    const int q_base = H / nks, q_rem = H - q_base * nks;
    int q_err = 0;
    int ds_t = 0, ms_t = 0;                               // k-step t
    int ds_r = (DP - 1) % DP, ms_r = (DP - 1) % MSP;      // k-step t + DP - 1 (rows issued this iteration)
    int ms_m = (2 * DP - 2) % MSP;                        // k-step t + 2 DP - 2 (metadata issued this iteration)
    // pacing: this workgroup's cohort = the workgroups of its XCD label in its dispatch generation (32 per XCD fit at one per
    // CU); sync point b sits at iteration ceil(b nks / blocks)
    const int pace_gen = (int)(blockIdx.x / kNumXcd) / 32;
    const int pace_members = pos_end - (pos0 + 32 * pace_gen) < 32 ? pos_end - (pos0 + 32 * pace_gen) : 32;
    int pace_b = 1, pace_next = a.pace ? (nks + a.pace_blocks - 1) / a.pace_blocks : 0x7FFFFFFF;
    for (int t = 0; t < nks; ++t) {
      if (a.pace && t == pace_next && blockIdx.y == 0 && pace_gen < kPaceGens) {   // workgroup-uniform
        if (wave == 0 && lane == 0) {
          int* const cnt = a.pace + ((xcd * kPaceGens + pace_gen) * kPaceBlocks + pace_b);
          __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int spin = 0; spin < 64; ++spin) {             // bounded: at most 64 polls, then go on regardless
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= pace_members) break;
            __builtin_amdgcn_s_sleep(8);
          }
        }
        ++pace_b;
        pace_next = (int)(((long long)pace_b * nks + a.pace_blocks - 1) / a.pace_blocks);
        if (pace_b >= a.pace_blocks || pace_b >= kPaceBlocks) pace_next = 0x7FFFFFFF;
      }
      // every panel operation of iteration t - (DP - 1) has landed: this wave's share of k-step t's rows and the
      // metadata of k-step t + DP - 1
      wait_vm(nops - mark_p[0]);
      __builtin_amdgcn_s_barrier();   // everybody's share of k-step t has landed; everybody is done reading k-step t - 1
      __builtin_amdgcn_sched_barrier(0);
      if (t + DP - 1 < nks) {         // workgroup-uniform
        issue_rows_p(ms_r, ds_r);     // into the slot k-step t - 1 has just left
        if (t + 2 * DP - 2 < nks) issue_meta_p(t + 2 * DP - 2, ms_m);
      }
#pragma unroll
      for (int i = 0; i < DP - 2; ++i) mark_p[i] = mark_p[i + 1];
      mark_p[DP - 2] = nops;

      // this iteration's stages: half of them before the k-step's matrix work, half after it, so that the wave comes back
      // to its residual ring twice per k-step (a ring slot can only be refilled when its stage is consumed)
      int q = q_base;
      q_err += q_rem;
      if (q_err >= nks) {
        q_err -= nks;
        ++q;
      }
      const int q_pre = (q + 1) >> 1;
      for (int s = 0; s < q_pre && h < H; ++s) resid_step();

      const unsigned mt = meta_p + (unsigned)ms_t * T::META_P_BYTES;
      const unsigned dt = data_p + (unsigned)ds_t * KSTEP_BYTES;
      ds_t = ds_t + 1 == DP ? 0 : ds_t + 1;
      ds_r = ds_r + 1 == DP ? 0 : ds_r + 1;
      ms_t = ms_t + 1 == MSP ? 0 : ms_t + 1;
      ms_r = ms_r + 1 == MSP ? 0 : ms_r + 1;
      ms_m = ms_m + 1 == MSP ? 0 : ms_m + 1;
      {
        const unsigned aw0 = lds_read_b32(mt + 4 * lane);            // row blocks 0 .. 3 (plan wave 2 wave)
        const unsigned aw1 = lds_read_b32(mt + 256 + 4 * lane);      // row blocks 4 .. 7 (plan wave 2 wave + 1)
        uint2_t blo[SLOTS], bhi[SLOTS];
        read_b_fragments(dt, blo, bhi);
        wait_lgkmcnt0();
        uint4_t bq[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) bq[s] = uint4_t{blo[s][0], blo[s][1], bhi[s][0], bhi[s][1]};
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const half8_t afrag = adjacency_to_half8_x2(j < 4 ? aw0 : aw1, 4 * (j & 3));
          mfma_block<T::BF16>(acc[j], afrag, bq);
        }
      }
      for (int s = q_pre; s < q && h < H; ++s) resid_step();
    }
  }
  // ======================================================= what is left of the residual (no barriers) ==============
  while (h < H) resid_step();
  wait_vmcnt<0>();  // nothing of this wave may still be writing LDS when the workgroup's LDS is released
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");   // the last MFMAs' results -> the epilogue's accumulator reads

  // ======================================================= epilogue: C written once, plain stores ====================
  const float oscale = kAScaleInv * (a.out_scale ? *a.out_scale : 1.0f);
  const int prow0 = panel * kFusedPanelRows + wave * (RB * 16) + 4 * (lane >> 4);
  const int ocol0 = fs0 + (lane & 15);
#pragma unroll
  for (int j = 0; j < RB; ++j) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = prow0 + 16 * j + i, col = ocol0 + 16 * s;
        if (col < F && row < a.num_nodes) a.output[(long long)row * F + col] = acc[j][s][i] * oscale;
      }
    }
  }
}

// Host launcher.  The plan arrays are padded as the builders pad them: panel_cols by 2 k-steps, panel_bits by one k-step,
// records by one record.  pace_blocks: sync points per column sweep between the workgroups of an XCD (0 / 1: none).
template <class T>
inline int launch_spmm_fused(const int* panel_ptr, const int* panel_cols, const uint32_t* panel_bits,
                             const int* panel_order, const int* wave_ptr, const uint32_t* records, int num_nodes,
                             int embedding_dim, const void* input, float* output, const float* out_scale,
                             hipStream_t stream, int pace_blocks = 0, const int* xcd_ptr = nullptr,
                             int max_panels_per_xcd = 0) {
  if (num_nodes < 0 || embedding_dim < 0) return kErrBadShape;
  if (num_nodes == 0 || embedding_dim == 0) return kOk;
  if (embedding_dim % 8 != 0 || ((uintptr_t)input & 15) || ((uintptr_t)records & 15)) return kErrBadShape;
  FusedArgs<T> a;
  a.panel_ptr = panel_ptr;
  a.panel_cols = panel_cols;
  a.panel_bits = panel_bits;
  a.panel_order = panel_order;
  a.wave_ptr = wave_ptr;
  a.records = records;
  a.input = static_cast<const typename FusedArgs<T>::in_t*>(input);
  a.output = output;
  a.out_scale = out_scale;
  a.num_nodes = num_nodes;
  a.num_panels = (num_nodes + kFusedPanelRows - 1) / kFusedPanelRows;
  a.panels_per_xcd = (a.num_panels + kNumXcd - 1) / kNumXcd;
  a.xcd_ptr = nullptr;
  if (xcd_ptr != nullptr) {
    if (max_panels_per_xcd < 1 || max_panels_per_xcd > a.num_panels) return kErrBadShape;
    a.xcd_ptr = xcd_ptr;
    a.panels_per_xcd = max_panels_per_xcd;
  }
  a.F = embedding_dim;
  const int slabs = (embedding_dim + T::FS - 1) / T::FS;
  a.meta_nt = slabs == 1;
  a.pace = nullptr;
  a.pace_blocks = 0;
  pace_blocks = pace_blocks < 0 ? 0 : (pace_blocks > kPaceBlocks ? kPaceBlocks : pace_blocks);
  if (pace_blocks > 1 && slabs == 1) {
    // one counter block per DEVICE (a process may drive several GPUs: a block allocated on the first one is foreign memory to the
    // others), found under a lock; never freed.  Only one paced launch per device may be in flight at a time (the counters are
    // cleared on the launch's own stream): pacing is an opt-in experiment (FUSED_PACE_BLOCKS = 0 by default), not a product path.
    constexpr int kMaxDevices = 64;
    static int* device_counters[kMaxDevices] = {};
    static std::mutex pace_mutex;
    const size_t bytes = sizeof(int) * kNumXcd * kPaceGens * kPaceBlocks;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return kErrLaunch;
    int* counters = nullptr;
    {
      std::lock_guard<std::mutex> lock(pace_mutex);
      if (device_counters[dev] == nullptr && hipMalloc(reinterpret_cast<void**>(&device_counters[dev]), bytes) != hipSuccess)
        return kErrLaunch;
      counters = device_counters[dev];
    }
    if (hipMemsetAsync(counters, 0, bytes, stream) != hipSuccess) return kErrLaunch;
    a.pace = counters;
    a.pace_blocks = pace_blocks;
  }
  const int lds_rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_fused_kernel<T>), T::BLOCK_LDS);
  if (lds_rc != kOk) return lds_rc;
  hipLaunchKernelGGL(spmm_fused_kernel<T>, dim3((unsigned)(a.panels_per_xcd * kNumXcd), (unsigned)slabs),
                     dim3(T::THREADS), T::BLOCK_LDS, stream, a);
  return hipGetLastError() == hipSuccess ? kOk : kErrLaunch;
}

}  // namespace voltrix
