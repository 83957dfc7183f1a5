// Voltrix-SpMM for MI355X (gfx950) -- the two-level format in ONE launch (round 3).
//
// spmm_panel_kernel + spmm_tc16_pair_kernel ran side by side on two streams and met in C through a zero fill and two
// float-atomic epilogues.  Here one 512-thread workgroup owns a 512-row panel for the whole
// product: every wave keeps the accumulators of its four 16-row blocks (128 registers) and feeds them from BOTH halves
// of the format,
//
//   shared columns    the panel plan (panel_ptr / panel_cols / panel_bits, spmm_panel_kernels.hpp): 32 rows of B gathered
//                     once per k-step into a ring shared by the workgroup (one s_barrier per k-step), 32 x
//                     v_mfma_f32_16x16x32_f16 per wave and k-step;
//   residual columns  the edges the plan left in the reference's window format, re-packed per wave into a stream of
//                     256-byte STAGE RECORDS (32 condensed columns of ONE of the wave's four windows: 32 rows of B, the
//                     16 bitmap words, the row block), the wave's four windows merged in column order so that its 64 rows
//                     sweep the sorted columns together.  A record is consumed in two half-stages of 16 gathered rows
//                     (4 KiB at FS = 128) through a wave-private ring of three half-slots, v_mfma_f32_16x16x16_f16, no
//                     barrier;
//
// interleaved: every k-step of the panel loop is followed by the wave's share of residual half-stages (spread evenly
// over the k-steps), so the matrix-core-bound and the gather-bound halves still overlap -- inside one wave now, not
// between two kernels -- and C is written ONCE with plain stores: no zero fill, no atomics, no second stream, no combine
// pass, and the result does not depend on any timing (one fixed summation order per row).
//
// vmcnt bookkeeping.  The two pipelines issue a data-dependent mix of LDS-DMAs, so no wait count is a compile-time
// constant.  Every wave counts the vector-memory operations it has issued (`nops`, a scalar) and remembers the count
// after each group it will wait for (`mark`); "that group has landed" is then s_waitcnt vmcnt(nops - mark), EXACT, picked
// from the 64 immediates by a scalar binary search (wait_vm): loads retire in issue order, so everything up to the mark is
// done as soon as at most nops - mark operations are outstanding.  Nothing else in the loop may issue vector memory
// operations (the ISA listing is checked for that: the loop holds only global_load_lds_* and MFMA / LDS / scalar code).
//
// LDS (FS = 128, DP = 3): panel ring 24 KiB + panel metadata 8 x 2.5 KiB + residual rings 8 x 12 KiB + residual
// metadata 8 x 1 KiB = 148 KiB: one workgroup per CU, two waves per SIMD with 256 registers each.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "voltrix/spmm_panel_kernels.hpp"

// Diagnostic builds only (-DVOLTRIX_EXPERIMENTAL, traits.hpp; harness/experiments/exp_fused_diag.py): bit 0 drops the residual half-steps, bit 1 the panel
// loop (the whole residual then runs barrier-free), bit 2 folds every residual row into the first 1024 rows of B (all L2
// hits).  Results are wrong by design; shipped kernels use 0.
#ifndef VOLTRIX_FUSED_DIAG
#define VOLTRIX_FUSED_DIAG 0
#endif

namespace voltrix {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef short short4_t __attribute__((ext_vector_type(4)));

constexpr int kFusedWaves = 8;          // waves per workgroup
constexpr int kFusedRowBlocks = 4;      // 16-row blocks (= windows) per wave
constexpr int kFusedPanelRows = kFusedWaves * kFusedRowBlocks * kBlkH;   // 512
constexpr int kRecordWords = 64;        // a residual stage record: 32 hind | 16 bitmap words | row block | pad
constexpr int kRecordBytes = 4 * kRecordWords;
constexpr int kRecordBlockWord = 48;    // word holding the record's row block (0 .. 3) inside its wave

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
  // ---- panel half (geometry of PanelTile<FS, DP, 8, 4, 1>) ----
  static constexpr int KSTEP_BYTES = kStageK * ROW_BYTES;
  static constexpr int NDMA_P = KSTEP_BYTES / 1024;                 // row DMAs per k-step, shared out over the waves
  static_assert(NDMA_P % WAVES == 0 || WAVES % NDMA_P == 0, "row DMAs per k-step vs waves");
  static constexpr int DPW = NDMA_P >= WAVES ? NDMA_P / WAVES : 1;  // per wave (surplus waves repeat the first ones)
  static constexpr int META_P_BYTES = 512;                          // 64 adjacency words + 64 column ids (32 used)
  static constexpr int MSP = 2 * DP - 1;                            // metadata slots per wave
  static constexpr int DATA_P = DP * KSTEP_BYTES;
  // ---- residual half: half-stages of 16 gathered rows ----
  static constexpr int DR = 3;                                      // half-slots in the wave-private ring
  static constexpr int HALF_BYTES = 16 * ROW_BYTES;
  static constexpr int NDMA_R = HALF_BYTES / 1024;
  static_assert(NDMA_R >= 1, "a half-stage is at least one 1-KiB DMA");
  static constexpr int MSR = 4;                                     // record slots per wave
  static constexpr int OFF_META_P = DATA_P;
  static constexpr int OFF_RING_R = OFF_META_P + WAVES * MSP * META_P_BYTES;
  static constexpr int OFF_META_R = OFF_RING_R + WAVES * DR * HALF_BYTES;
  static constexpr int BLOCK_LDS = OFF_META_R + WAVES * MSR * kRecordBytes;
  static_assert(BLOCK_LDS <= 160 * 1024, "LDS per CU");
};

// Residual half-slot image: 16 rows; a transposed read touches, per 32-lane half, the 8 rows 8y .. 8y + 7 (lane group g
// reads rows 4g .. 4g + 3).  Logical 32-byte slot s of row r lives at physical slot s ^ half_swizzle(r): the 8 rows of a
// half land on distinct bank groups for every FS.
template <int SLOTS>
__device__ __forceinline__ constexpr int half_swizzle(int r) {
  return SLOTS >= 8 ? (r & 7) : (SLOTS == 4 ? ((r >> 1) & 3) : ((r >> 2) & 1));
}

// one adjacency nibble (4 condensed columns of one row) -> two packed fp16x2 registers holding 2.0 / 0.0
__device__ __forceinline__ half4_t nibble_to_half4_x2(const unsigned n) {
  unsigned z;
  asm("v_lshl_or_b32 %0, %1, 15, %1" : "=v"(z) : "v"(n));
  uint2_t r;
  r[0] = (z << 14) & 0x40004000u;  // columns 0, 1
  r[1] = (z << 12) & 0x40004000u;  // columns 2, 3
  return __builtin_bit_cast(half4_t, r);
}

// acc[j][s] += A x B[s] for NS consecutive 16-column slots, j a wave-uniform run-time value in 0 .. 3, on
// v_mfma_f32_16x16x16_f16 (_bf16).  The accumulators are registers and the row block is data, so one of four register
// sets has to be picked at run time.  A C++ switch around the builtin made hipcc split the accumulators' live ranges at
// the join: 128 accumulator registers copied per half-step (1.3 us per half-step and wave instead of 0.2).  Here the
// four-way scalar branch sits INSIDE one asm statement that ties all four candidate sets in place ("+v"): no control
// flow the register allocator can see, no copies.  Hazards: the MFMAs of a statement touch different accumulators; the
// A / B operands come from VALU / LDS results the compiler has already waited for; the next reader of an accumulator
// (next k-step, a later half-step, the epilogue) is hundreds of cycles and at least one s_waitcnt away.
#define VOLTRIX_MFMA16(OP, C, B) OP " %" #C ", %[a], %[" #B "], %" #C "\n"
template <bool BF16>
__device__ __forceinline__ void mfma16_select4(float4_t (&c0)[4], float4_t (&c1)[4], float4_t (&c2)[4], float4_t (&c3)[4],
                                               const half4_t a, const uint2_t (&b)[4], const int j) {
#define VOLTRIX_MFMA16_BODY(OP)                                                                                       \
  asm volatile("s_cmp_lt_u32 %[j], 2\n s_cbranch_scc1 2f\n s_cmp_eq_u32 %[j], 2\n s_cbranch_scc1 1f\n"              \
               VOLTRIX_MFMA16(OP, 12, b0) VOLTRIX_MFMA16(OP, 13, b1) VOLTRIX_MFMA16(OP, 14, b2) VOLTRIX_MFMA16(OP, 15, b3) \
               "s_branch 9f\n1:\n"                                                                                   \
               VOLTRIX_MFMA16(OP, 8, b0) VOLTRIX_MFMA16(OP, 9, b1) VOLTRIX_MFMA16(OP, 10, b2) VOLTRIX_MFMA16(OP, 11, b3)   \
               "s_branch 9f\n2:\n s_cmp_eq_u32 %[j], 0\n s_cbranch_scc1 3f\n"                                        \
               VOLTRIX_MFMA16(OP, 4, b0) VOLTRIX_MFMA16(OP, 5, b1) VOLTRIX_MFMA16(OP, 6, b2) VOLTRIX_MFMA16(OP, 7, b3)     \
               "s_branch 9f\n3:\n"                                                                                   \
               VOLTRIX_MFMA16(OP, 0, b0) VOLTRIX_MFMA16(OP, 1, b1) VOLTRIX_MFMA16(OP, 2, b2) VOLTRIX_MFMA16(OP, 3, b3)     \
               "9:\n"                                                                                                \
               : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]), \
                 "+v"(c2[0]), "+v"(c2[1]), "+v"(c2[2]), "+v"(c2[3]), "+v"(c3[0]), "+v"(c3[1]), "+v"(c3[2]), "+v"(c3[3]) \
               : [a] "v"(a), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [j] "s"(j)               \
               : "scc")
  if constexpr (BF16)
    VOLTRIX_MFMA16_BODY("v_mfma_f32_16x16x16_bf16");
  else
    VOLTRIX_MFMA16_BODY("v_mfma_f32_16x16x16_f16");
#undef VOLTRIX_MFMA16_BODY
}
#undef VOLTRIX_MFMA16

template <class T>
struct FusedArgs {
  using in_t = typename std::conditional<T::BF16, bfloat16_bits, _Float16>::type;
  const int* panel_ptr;        // [NP+1]
  const int* panel_cols;       // [32 * (S + 2)]
  const uint32_t* panel_bits;  // [(S + 1) * 8 * 64]
  const int* panel_order;      // optional: launch position -> panel (longest first); nullptr = natural
  const int* wave_ptr;         // [8 NP + 1]: first residual stage record of (panel, wave)
  const uint32_t* records;     // [R + 1][64]
  const in_t* input;
  float* output;
  const float* out_scale;      // optional device scalar (SpmmArgs::out_scale)
  int num_nodes;
  int num_panels;
  int panels_per_xcd;
  int F;
  int meta_nt;                 // 1: metadata DMAs are non-temporal (one slab covers F: every byte is read once)
  int* pace;                   // EXPERIMENT (VOLTRIX_FUSED_PACE, exp_fused_pace.py): zeroed int32 [8][kPaceGens][kPaceBlocks]
                               // arrival counters, or nullptr (shipped).  The workgroups that share an XCD and a dispatch
                               // generation wait for each other -- bounded, advisory: correctness never depends on it -- at
                               // pace_blocks points of their column sweep, so that an XCD's 16 k resident rows sweep the
                               // columns together (the L2-hit lever of DESIGN.md section 3.7)
  int pace_blocks;
};
constexpr int kPaceGens = 8, kPaceBlocks = 64;

template <class T>
static __global__ __launch_bounds__(T::THREADS) void spmm_fused_kernel(const FusedArgs<T> a) {
  constexpr int FS = T::FS, DP = T::DP, MSP = T::MSP, DR = T::DR, MSR = T::MSR, RB = T::RB;
  constexpr int ROW_BYTES = T::ROW_BYTES, KSTEP_BYTES = T::KSTEP_BYTES, DPW = T::DPW, HALF_BYTES = T::HALF_BYTES;
  constexpr int RPD = T::ROWS_PER_DMA, LPR = T::LANES_PER_ROW, SLOTS = T::SLOTS, NDMA_R = T::NDMA_R;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int lane = threadIdx.x & (kWave - 1);

  // XCD x = blockIdx.x % 8 owns a contiguous range of launch positions (as spmm_panel_kernel)
  const int xcd = blockIdx.x % kNumXcd;
  const int pos = xcd * a.panels_per_xcd + (int)(blockIdx.x / kNumXcd);
  const int pos_end = (xcd + 1) * a.panels_per_xcd < a.num_panels ? (xcd + 1) * a.panels_per_xcd : a.num_panels;
  if (pos >= pos_end) return;  // workgroup-uniform
  const int panel = a.panel_order ? a.panel_order[pos] : pos;
  const int fs0 = blockIdx.y * FS;
  const int F = a.F;

  const int ks0 = a.panel_ptr[panel];
  const int nks = (VOLTRIX_FUSED_DIAG & 2) ? 0 : a.panel_ptr[panel + 1] - ks0;
  const int rec0 = a.wave_ptr[panel * T::WAVES + wave];
  const int nrec = (VOLTRIX_FUSED_DIAG & 1) ? 0 : a.wave_ptr[panel * T::WAVES + wave + 1] - rec0;
  const int H = 2 * nrec;      // residual half-steps of this wave

  float4_t acc[RB][SLOTS];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) acc[j][s] = float4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned data_p = lds0;
  const unsigned meta_p = lds0 + T::OFF_META_P + (unsigned)wave * (MSP * T::META_P_BYTES);
  const unsigned ring_r = lds0 + T::OFF_RING_R + (unsigned)wave * (DR * HALF_BYTES);
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
    hr_off_p[d] = 256 + 4 * r;
  }
  const uint32_t* const bits_base = a.panel_bits + ((long long)ks0 * T::WAVES + wave) * kWave;
  const int* const cols_base = a.panel_cols + (long long)ks0 * kStageK;
  auto issue_meta_p = [&](const int s, const int ms) {   // k-step s < nks: 2 operations
    const unsigned dst = meta_p + (unsigned)ms * T::META_P_BYTES;
    int ml = lane;
    asm volatile("" : "+v"(ml));
    if (a.meta_nt) {
      dma_b32_nt(bits_base + (long long)s * (T::WAVES * kWave) + ml, dst);
      dma_b32_nt(cols_base + (long long)s * kStageK + ml, dst + 256);
    } else {
      dma_b32(bits_base + (long long)s * (T::WAVES * kWave) + ml, dst);
      dma_b32(cols_base + (long long)s * kStageK + ml, dst + 256);
    }
    nops += 2;
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
  // transposed B reads of the panel ring (as spmm_panel_kernel)
  const int trow_p = 8 * g + q4;
  const int tr_z_p = slot_swizzle<SLOTS>(trow_p);
  unsigned rd_off_p = trow_p * ROW_BYTES + 8 * p4 + (tr_z_p << 5);
  int tr_delta_p[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tr_delta_p[b] = ((tr_z_p >> b) & 1) ? -(32 << b) : (32 << b);
    asm volatile("" : "+v"(tr_delta_p[b]));
  }
  asm volatile("" : "+v"(rd_off_p));

  // ======================================================= residual half: lane constants, issue helpers ===========
  // row DMA i of a half-stage writes LDS rows i * RPD + lane / LPR; the lane's 16 bytes come from column chunk
  // (lane % LPR) of that row of B, slot-swizzled on the source
  const char* cbase_r[NDMA_R];
#pragma unroll
  for (int i = 0; i < NDMA_R; ++i) {
    const int r = i * RPD + lane / LPR;
    const int c = lane % LPR;
    int col = fs0 + (((c >> 1) ^ half_swizzle<SLOTS>(r)) * 16) + (c & 1) * 8;
    col = col < F ? col : fs0;
    unsigned long long cb = (unsigned long long)((const char*)a.input + (long long)col * 2);
    asm volatile("" : "+v"(cb));
    cbase_r[i] = (const char*)cb;
  }
  const unsigned id_off_r = 4 * (lane / LPR);   // + 64 * half + 4 * RPD * i: the lane's row of DMA i inside the record
  // A fragment of v_mfma_f32_16x16x16: lane (R, g) holds row R, condensed columns 4g .. 4g + 3 of the half-stage =
  // TC block 2 half + (g >> 1), columns 4 (g & 1) ..: nibble R & 7 of bitmap word 4 block + (R >> 3) + 2 (g & 1)
  const unsigned aw_off_r = 128 + 4 * (4 * (g >> 1) + (R >> 3) + 2 * (g & 1));   // + 32 * half
  const unsigned a_shift = 4 * (R & 7);
  // transposed B reads: lane (g, q, p) supplies row 4g + q, bytes 8p .. of logical slot s
  const int trow_r = 4 * g + q4;
  const int tr_z_r = half_swizzle<SLOTS>(trow_r);
  unsigned rd_off_r = trow_r * ROW_BYTES + 8 * p4 + (tr_z_r << 5);
  int tr_delta_r[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tr_delta_r[b] = ((tr_z_r >> b) & 1) ? -(32 << b) : (32 << b);
    asm volatile("" : "+v"(tr_delta_r[b]));
  }
  asm volatile("" : "+v"(rd_off_r));

  const uint32_t* const rec_base = a.records + (long long)rec0 * kRecordWords;
  auto issue_record = [&](const int s, const int ms) {    // record s < nrec: 1 operation
    int ml = lane;
    asm volatile("" : "+v"(ml));
    const uint32_t* const src = rec_base + (long long)s * kRecordWords + ml;
    if (a.meta_nt)
      dma_b32_nt(src, meta_r + (unsigned)ms * kRecordBytes);
    else
      dma_b32(src, meta_r + (unsigned)ms * kRecordBytes);
    nops += 1;
  };
  // rows of half-stage (record in slot ms, half hh) into ring slot rs; ids[] = the lane's rows, read from the record
  auto read_ids = [&](const int ms, const int hh, unsigned (&ids)[NDMA_R]) {
    const unsigned base = meta_r + (unsigned)ms * kRecordBytes + id_off_r + 64u * (unsigned)hh;
#pragma unroll
    for (int i = 0; i < NDMA_R; ++i) ids[i] = lds_read_b32(base + 4 * RPD * i);
  };
  auto fold_ids = [&](unsigned (&ids)[NDMA_R]) {
    if (VOLTRIX_FUSED_DIAG & 4) {
#pragma unroll
      for (int i = 0; i < NDMA_R; ++i) ids[i] &= 1023u;
    }
  };
  auto issue_half = [&](const int rs, const unsigned (&ids)[NDMA_R]) {   // NDMA_R operations
    const unsigned dst = ring_r + (unsigned)rs * HALF_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA_R; ++i) dma_b128(cbase_r[i] + (unsigned long long)ids[i] * row_bytes, dst + i * 1024);
    nops += NDMA_R;
  };

  // ======================================================= prologue ===============================================
  // panel: metadata of k-steps 0 .. DP-2; residual: records 0 .. 2; then the first rows of both rings
#pragma unroll
  for (int s = 0; s < DP - 1; ++s)
    if (s < nks) issue_meta_p(s, s % MSP);
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < nrec) issue_record(s, s % MSR);
  wait_vmcnt<0>();
  __builtin_amdgcn_sched_barrier(0);

  int mark_p[DP - 1];   // mark_p[i]: nops after the panel issues of the iteration that is DP-1-i iterations back
#pragma unroll
  for (int s = 0; s < DP - 1; ++s) {
    if (s < nks) issue_rows_p(s % MSP, s % DP);
    if (s + DP - 1 < nks) issue_meta_p(s + DP - 1, (s + DP - 1) % MSP);
    mark_p[s] = nops;
  }
  int mark_r[DR];       // mark_r[i]: nops after the refill that filled the half-stage consumed i steps from now
#pragma unroll
  for (int h0 = 0; h0 < DR; ++h0) {
    if (h0 < H) {
      unsigned ids[NDMA_R];
      read_ids((h0 >> 1) % MSR, h0 & 1, ids);
      wait_lgkmcnt0();
      fold_ids(ids);
      issue_half(h0, ids);
    }
    mark_r[h0] = nops;
  }

  // ======================================================= one residual half-step =================================
  int h = 0;            // next half-step
  int rs_h = 0;         // h % DR
  int ms_h = 0;         // (h >> 1) % MSR
  int hh_h = 0;         // h & 1
  auto resid_step = [&]() {
    wait_vm(nops - mark_r[0]);                 // rows of half-stage h (and every record fetched before them)
    const unsigned rec = meta_r + (unsigned)ms_h * kRecordBytes;
    const unsigned jw = lds_read_b32(rec + 4 * kRecordBlockWord);
    const unsigned aword = lds_read_b32(rec + aw_off_r + 32u * (unsigned)hh_h);
    const bool refill = h + DR < H;            // wave-uniform
    // half-stage h + 3: record (h + 3) >> 1 = s + 1 (h even) or s + 2 (h odd), half (h + 3) & 1
    const int ms_n = (ms_h + 1 + hh_h) % MSR;
    unsigned ids[NDMA_R];
    if (refill) read_ids(ms_n, hh_h ^ 1, ids);
    unsigned taddr[SLOTS];
    taddr[0] = ring_r + (unsigned)rs_h * HALF_BYTES + rd_off_r;
#pragma unroll
    for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
      for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta_r[b];
    uint2_t bf[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) bf[s] = lds_read_tr16_b64<0>(taddr[s]);
    wait_lgkmcnt0();
    // the record three ahead (once per record, on its first half), then the rows of half-stage h + 3 into the slot just read
    if (hh_h == 0 && (h >> 1) + 3 < nrec) issue_record((h >> 1) + 3, (ms_h + 3) % MSR);
    if (refill) {
      fold_ids(ids);
      issue_half(rs_h, ids);
    }
#pragma unroll
    for (int i = 0; i < DR - 1; ++i) mark_r[i] = mark_r[i + 1];
    mark_r[DR - 1] = nops;

    const half4_t afrag = nibble_to_half4_x2((aword >> a_shift) & 0xFu);
    const int j = __builtin_amdgcn_readfirstlane((int)jw) & 3;
    // The row block is run-time data, the accumulators are registers: a wave-uniform switch picks the set.  The MFMAs are
    // inline asm with the accumulator tied in place ("+v"): with the builtin, hipcc wrote each case's results to fresh
    // registers and re-joined the four paths by copying all 128 accumulator registers every half-step (1.3 us per
    // half-step and wave instead of 0.2).  Hazards: the eight MFMAs of a case touch eight different accumulators, and the
    // next reader of any of them (the next k-step's MFMAs, a later half-step, the epilogue) is hundreds of cycles away,
    // behind at least one s_waitcnt -- no software wait states are needed around the block.
    // slots in groups of four (FS = 32: the two slots twice -- the second pair of accumulators / fragments is a scratch copy)
    if constexpr (SLOTS >= 4) {
#pragma unroll
      for (int s0 = 0; s0 < SLOTS; s0 += 4) {
        float4_t(&c0)[4] = reinterpret_cast<float4_t(&)[4]>(acc[0][s0]);
        float4_t(&c1)[4] = reinterpret_cast<float4_t(&)[4]>(acc[1][s0]);
        float4_t(&c2)[4] = reinterpret_cast<float4_t(&)[4]>(acc[2][s0]);
        float4_t(&c3)[4] = reinterpret_cast<float4_t(&)[4]>(acc[3][s0]);
        const uint2_t(&bq)[4] = reinterpret_cast<const uint2_t(&)[4]>(bf[s0]);
        mfma16_select4<T::BF16>(c0, c1, c2, c3, afrag, bq, j);
      }
    } else {
      float4_t c[4][4];
      uint2_t bq[4] = {bf[0], bf[1], bf[0], bf[1]};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        c[jj][0] = acc[jj][0];
        c[jj][1] = acc[jj][1];
        c[jj][2] = c[jj][3] = float4_t{0.f, 0.f, 0.f, 0.f};
      }
      mfma16_select4<T::BF16>(c[0], c[1], c[2], c[3], afrag, bq, j);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        acc[jj][0] = c[jj][0];
        acc[jj][1] = c[jj][1];
      }
    }
    ++h;
    rs_h = rs_h + 1 == DR ? 0 : rs_h + 1;
    ms_h = hh_h ? (ms_h + 1 == MSR ? 0 : ms_h + 1) : ms_h;
    hh_h ^= 1;
  };

  // ======================================================= panel loop: one k-step + this wave's share of half-steps ==
  if (nks > 0) {
    // half-steps spread evenly over the k-steps: q_t = base (+ 1 whenever the remainder accumulator wraps)
    const int q_base = H / nks, q_rem = H - q_base * nks;
    int q_err = 0;
    int ds_t = 0, ms_t = 0;                               // k-step t
    int ds_r = (DP - 1) % DP, ms_r = (DP - 1) % MSP;      // k-step t + DP - 1 (rows issued this iteration)
    int ms_m = (2 * DP - 2) % MSP;                        // k-step t + 2 DP - 2 (metadata issued this iteration)
    // pacing (experiment): this workgroup's cohort = the workgroups of its XCD label in its dispatch generation (32 per XCD
    // fit at one per CU); sync point b sits at iteration ceil(b nks / blocks)
    const int pace_gen = (int)(blockIdx.x / kNumXcd) / 32;
    const int pace_members = pos_end - (xcd * a.panels_per_xcd + 32 * pace_gen) < 32
                                 ? pos_end - (xcd * a.panels_per_xcd + 32 * pace_gen) : 32;
    int pace_b = 1, pace_next = a.pace ? (nks + a.pace_blocks - 1) / a.pace_blocks : 0x7FFFFFFF;
    for (int t = 0; t < nks; ++t) {
      if (a.pace && t == pace_next && blockIdx.y == 0 && pace_gen < kPaceGens) {   // workgroup-uniform
        if (wave == 0 && lane == 0) {
          int* const cnt = a.pace + ((xcd * kPaceGens + pace_gen) * kPaceBlocks + pace_b);
          __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int spin = 0; spin < 64; ++spin) {             // bounded: at most 64 polls (~0.1 ms), then go on regardless
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

      // this iteration's half-steps: half of them before the k-step's matrix work, half after it, so that the wave comes
      // back to its residual ring twice per k-step (a ring slot can only be refilled when its half-stage is consumed; all
      // half-steps behind the k-step left the ring idle for most of the iteration: 2.67 ms against 1.35 ms for the pair)
      int q = q_base;
      q_err += q_rem;
      if (q_err >= nks) {
        q_err -= nks;
        ++q;
      }
      const int q_pre = (q + 1) >> 1;
      for (int s = 0; s < q_pre && h < H; ++s) resid_step();

      const unsigned mt = meta_p + (unsigned)ms_t * T::META_P_BYTES;
      const unsigned dt = data_p + (unsigned)ds_t * KSTEP_BYTES + rd_off_p;
      ds_t = ds_t + 1 == DP ? 0 : ds_t + 1;
      ds_r = ds_r + 1 == DP ? 0 : ds_r + 1;
      ms_t = ms_t + 1 == MSP ? 0 : ms_t + 1;
      ms_r = ms_r + 1 == MSP ? 0 : ms_r + 1;
      ms_m = ms_m + 1 == MSP ? 0 : ms_m + 1;
      {
        const unsigned aw = lds_read_b32(mt + 4 * lane);
        unsigned taddr[SLOTS];
        taddr[0] = dt;
#pragma unroll
        for (int b = 0; (1 << b) < SLOTS; ++b)
#pragma unroll
          for (int s = (1 << b); s < (2 << b) && s < SLOTS; ++s) taddr[s] = taddr[s - (1 << b)] + tr_delta_p[b];
        uint2_t blo[SLOTS], bhi[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
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
            if constexpr (T::BF16)
              acc[j][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afrag),
                                                                  __builtin_bit_cast(bf16x8_t, bq), acc[j][s], 0, 0, 0);
            else
              acc[j][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, __builtin_bit_cast(half8_t, bq), acc[j][s], 0, 0, 0);
          }
        }
      }
      for (int s = q_pre; s < q && h < H; ++s) resid_step();
    }
  }
  // ======================================================= what is left of the residual (no barriers) ==============
  while (h < H) resid_step();
  wait_vmcnt<0>();  // nothing of this wave may still be writing LDS when the workgroup's LDS is released

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
// records by one record.
template <class T>
inline int launch_spmm_fused(const int* panel_ptr, const int* panel_cols, const uint32_t* panel_bits,
                             const int* panel_order, const int* wave_ptr, const uint32_t* records, int num_nodes,
                             int embedding_dim, const void* input, float* output, const float* out_scale,
                             hipStream_t stream, int pace_blocks = 0 /* sync points per column sweep (0: none) */) {
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
  a.F = embedding_dim;
  const int slabs = (embedding_dim + T::FS - 1) / T::FS;
  a.meta_nt = slabs == 1;
  a.pace = nullptr;
  a.pace_blocks = 0;
  pace_blocks = pace_blocks < 0 ? 0 : (pace_blocks > kPaceBlocks ? kPaceBlocks : pace_blocks);
  if (pace_blocks > 1 && slabs == 1) {
    static int* counters = nullptr;
    const size_t bytes = sizeof(int) * kNumXcd * kPaceGens * kPaceBlocks;
    if (counters == nullptr && hipMalloc(reinterpret_cast<void**>(&counters), bytes) != hipSuccess) return kErrLaunch;
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
